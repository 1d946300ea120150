#!/usr/bin/env python3
"""Randomised sweep of the sliding-window engine (k_window_ticks) against the refit-per-tick oracle (test
infrastructure: uses oracle/): random window length N (around the 16-column panel boundaries), input dimension, kernel,
stream length (several ring compactions), number of independent windows and block cuts of the stream.
   python tests/fuzz/fuzz_window.py [seconds=60] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NS = [2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 100, 129]
t_end, cases, bad, worst = time.time() + budget, 0, 0, 0.0
while time.time() < t_end:
    N = int(rng.choice(NS))
    kid = int(rng.integers(0, 3))
    d = 1 if kid == synth.KERNEL_RBF_BROWNIAN else int(rng.integers(1, 5))
    T = int(rng.integers(max(2, N // 2), 3 * N + 20))
    nwin = int(rng.integers(1, 4))
    if N >= 64 and rng.integers(0, 3) == 0:
        nwin = 512              # from 512 windows the steady-state ticks go four per pass (k_window_multi): three of the windows are compared
    noise = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1 << 30))
    Xs, ys = [], []
    for w in range(nwin):
        r2 = np.random.default_rng(seed + w)
        t = np.arange(11 + w, 11 + w + T, dtype=np.float64)
        y = synth._slip_series(r2, t)
        X = t[:, None] if d == 1 else np.column_stack([(t - t.mean()) / t.std()] + [r2.normal(size=T) for _ in range(d - 1)])
        Xs.append(X); ys.append(y)
    X, y = np.stack(Xs), np.stack(ys)
    theta = {2: np.array([0.5, 30.0, 0.01, 0.002]), 0: np.array([0.02, 1.0, 1e-3]),
             1: np.concatenate([[0.02], np.linspace(0.8, 1.6, d), [1e-3]])}[kid]
    ctx = engine.Context(max_n=8, max_m=8, max_d=d)
    ctx.window_init(nwin, N, d, kid, theta)
    cuts = sorted(set([0, T] + [int(c) for c in rng.integers(1, T, size=int(rng.integers(0, 4)))]))
    outs = [ctx.window_push(X[:, a:b], y[:, a:b], include_noise=noise) for a, b in zip(cuts[:-1], cuts[1:])]
    pm, pv, lm = [np.concatenate([o[i] for o in outs], axis=1) for i in range(3)]
    cases += 1
    tag = f"N={N} d={d} kid={kid} T={T} nwin={nwin} noise={noise} cuts={cuts} seed={seed}"
    for w in (range(nwin) if nwin <= 4 else sorted({0, nwin - 1, int(rng.integers(0, nwin))})):
        opm, opv, olm = go.sliding_window_stream(kid, theta, N, X[w], y[w], include_noise=noise)
        scale = np.abs(opv) if noise else np.maximum(np.abs(opv), 1e-9 * go.kernel_Kdiag(kid, theta, X[w]))
        e = max(float(np.max(np.abs(pm[w] - opm)) / max(np.max(np.abs(opm)), 1e-12)), float(np.max(np.abs(pv[w] - opv) / scale)),
                float(np.max(np.abs(lm[w] - olm) / np.maximum(np.abs(olm), 1.0))))
        worst = max(worst, e / 1e-6)
        if not (e < 1e-6) or ctx.window_state(w) != (min(N, T), 0):
            print("FAIL", tag, "window", w, "err", e, "state", ctx.window_state(w)); bad += 1
print(f"cases {cases} failures {bad} worst error / bar {worst:.3g}")
sys.exit(1 if bad else 0)
