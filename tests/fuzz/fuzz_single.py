#!/usr/bin/env python3
"""Randomised sweep of the single-window entry points against the oracle (test infrastructure: uses oracle/):
cgp_fit -> cgp_predict (with and without the noise term) -> cgp_get_alpha -> cgp_get_factor -> cgp_nll_grad, fp64,
random N around the tile boundaries, random M, d and kernel.
   python tests/fuzz/fuzz_single.py [seconds=60] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NS = [3, 16, 17, 127, 128, 129, 134, 255, 256, 257, 300, 384, 385, 512, 513, 700]
MS = [1, 5, 127, 128, 129, 300, 599]
t_end, cases, bad, worst = time.time() + budget, 0, 0, 0.0
ctx = engine.Context(max_n=max(NS), max_m=max(max(MS), max(NS)), max_d=6)


def check(name, err, bar, tag):
    global bad, worst
    worst = max(worst, err / bar)
    if not (err < bar):
        print("FAIL", name, tag, "err", err, "bar", bar)
        bad += 1


while time.time() < t_end:
    N, M = int(rng.choice(NS)), int(rng.choice(MS))
    kid = int(rng.integers(0, 3))
    d = 1 if kid == synth.KERNEL_RBF_BROWNIAN else int(rng.integers(1, 7))
    seed = int(rng.integers(0, 1 << 30))
    X, y, Xs = synth.window(N, d, M, seed)
    if kid == synth.KERNEL_RBF_BROWNIAN:
        X = (np.arange(N, dtype=np.float64) + 11.0 + (seed % 50))[:, None]
        Xs = (X[-1, 0] + 1.0 + np.arange(M, dtype=np.float64))[:, None]
    th = synth.theta_for(kid, d, y, np.random.default_rng(seed + 7) if kid == synth.KERNEL_SE_ARD else None)
    tag = f"N={N} M={M} d={d} kid={kid} seed={seed}"
    f = go.fit(kid, th, X, y)
    rc, logml = ctx.fit(X, y, kid, th)
    cases += 1
    if rc != 0:
        print("FAIL fit rc", tag, rc); bad += 1
        continue
    if f.jitter > 0:
        continue
    check("logml", abs(logml - f.logml) / abs(f.logml), 1e-6, tag)
    noise = bool(rng.integers(0, 2))
    mean, var = ctx.predict(Xs, include_noise=noise)
    omu, ovar = go.predict(f, Xs, include_noise=noise)
    check("mean", float(np.max(np.abs(mean - omu)) / max(np.max(np.abs(omu)), 1e-300)), 1e-6, tag)
    # without the noise term the variance may sit at GPy's 1e-15 floor: compare on the scale of the prior variance
    scale = np.abs(ovar) if noise else np.maximum(np.abs(ovar), 1e-9 * go.kernel_Kdiag(kid, th, Xs))
    check("var", float(np.max(np.abs(var - ovar) / scale)), 1e-6, tag)
    Ky = go.kernel_K(kid, th, X) + (go.noise_var(kid, th) + 1e-8) * np.eye(N)
    a = ctx.alpha()
    check("alpha", float(np.max(np.abs(Ky @ a - y)) / np.max(np.abs(y))), 1e-6, tag)
    L = ctx.factor()
    check("factor", float(np.max(np.abs(np.tril(L) @ np.tril(L).T - Ky)) / np.max(np.abs(Ky))), 1e-12, tag)
    if N <= 513:
        nll, g = ctx.nll_grad(X, y, kid, th)
        onll, og = go.nll_and_grad(kid, th, X, y)
        check("nll", abs(nll - onll) / abs(onll), 1e-6, tag)
        check("grad", float(np.max(np.abs(g - og)) / np.max(np.abs(og))), 1e-6, tag)
print(f"cases {cases} failures {bad} worst error / bar {worst:.3g}")
sys.exit(1 if bad else 0)
