import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from corenav_gp_amd import engine
def run(nwin, N, d, kid, T2):
    rng = np.random.default_rng(5)
    T = N + 8 + T2
    t = np.arange(11, 11 + T, dtype=np.float64)
    if d == 1: X = np.repeat(t[None, :, None], nwin, 0)
    else:
        X = np.empty((nwin, T, d)); X[:, :, 0] = (t - t.mean()) / t.std(); X[:, :, 1:] = rng.normal(size=(nwin, T, d - 1))
    y = 0.1 * np.sin(2 * np.pi * t / 40.0)[None] + rng.normal(0, 0.03, (nwin, T))
    theta = {2: np.array([0.5, 30.0, 0.01, 0.002]), 1: np.concatenate([[0.02], np.linspace(0.8, 1.6, d), [1e-3]])}[kid]
    A = engine.Context(max_n=8, max_m=8, max_d=d); A.window_init(nwin, N, d, kid, theta)
    B = engine.Context(max_n=8, max_m=8, max_d=d); B.window_init(nwin, N, d, kid, theta)
    n0 = N + 8
    for i in range(n0):
        A.window_push(X[:, i:i + 1], y[:, i:i + 1]); B.window_push(X[:, i:i + 1], y[:, i:i + 1])
    try:
        pa = A.window_push(X[:, n0:], y[:, n0:])
    except Exception as e:
        print(f"nwin {nwin} N {N} d {d} kid {kid} T2 {T2}: push failed {e}"); return
    pb = [np.concatenate(z, 1) for z in zip(*[B.window_push(X[:, i:i + 1], y[:, i:i + 1]) for i in range(n0, T)])]
    for name, a, b in zip(("mean", "var", "logml"), pa, pb):
        e = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
        bad = np.argwhere(~(e < 1e-8))
        print(f"nwin {nwin} N {N} d {d} kid {kid} T2 {T2}: {name} max rel diff {np.nanmax(e):.2e} nan {np.isnan(a).sum()}" + (f" first bad (win, tick) {bad[0]} a {a[tuple(bad[0])]} b {b[tuple(bad[0])]}" if len(bad) else ""))
for args in [(1, 64, 3, 1, 4), (1, 64, 3, 1, 24), (1, 80, 1, 2, 17), (1, 100, 6, 1, 40), (2, 512, 3, 1, 40), (300, 67, 3, 1, 13), (3, 70, 2, 1, 23)]:
    run(*args)
