#!/usr/bin/env python3
"""Randomised sweep of the one-launch short-window kernel (csrc/cgp_small.hpp; test infrastructure: uses oracle/): every
window length 2 ... 160, d = 1 ... 8, the three kernels, through cgp_nll_grad (value, gradient, jitter of GPy's ladder
against the oracle), cgp_predict after it (the lazy refit), cgp_fit_predict_batch on the same window (fit and predictions in
one launch, k_small_predict, a random number of test points, in a batch slot among others), and -- every few cases -- cgp_optimize against
cgp_optimize_batch on the same window (both run the device L-BFGS; they must agree to rounding) and against the
objective re-evaluated by the oracle at the returned optimum.
   python tests/fuzz/fuzz_small.py [seconds=60] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, cases, bad, worst, nopt, njit, nedge = time.time() + budget, 0, 0, 0.0, 0, 0, 0
ctx = engine.Context(max_n=160, max_m=160, max_d=8)
ctxb = engine.Context(max_n=160, max_m=160, max_d=8, max_batch=3)
ctxp = engine.Context(max_n=160, max_m=160, max_d=8, max_batch=5)
npred = 0


def check(name, err, bar, tag):
    global bad, worst
    worst = max(worst, err / bar)
    if not (err < bar):
        print("FAIL", name, tag, "err", err, "bar", bar)
        bad += 1


while time.time() < t_end:
    N = int(rng.integers(2, 161))
    kid = int(rng.integers(0, 3))
    d = 1 if kid == synth.KERNEL_RBF_BROWNIAN else int(rng.integers(1, 9))
    seed = int(rng.integers(0, 1 << 30))
    X, y, Xs = synth.window(N, d, 7, seed)
    if kid == synth.KERNEL_RBF_BROWNIAN:
        X = (np.arange(N, dtype=np.float64) + 11.0 + (seed % 5000))[:, None]
        Xs = (X[-1, 0] + 1.0 + np.arange(7, dtype=np.float64))[:, None]
    th = synth.theta_for(kid, d, y, np.random.default_rng(seed + 7) if kid == synth.KERNEL_SE_ARD else None)
    if rng.integers(0, 12) == 0 and N >= 6 and kid != synth.KERNEL_RBF_BROWNIAN:   # a window that needs GPy's jitter ladder
        X[1::2] = X[0:-1:2][: len(X[1::2])]
        th = th.copy()
        th[0], th[-1] = 1e9, 1e-10
    tag = f"N={N} d={d} kid={kid} seed={seed}"
    cases += 1
    try:
        f = go.fit(kid, th, X, y)
    except Exception:
        continue                                      # not positive definite even with the ladder: nothing to compare
    nll, g = ctx.nll_grad(X, y, kid, th)
    onll, og = go.nll_and_grad(kid, th, X, y)
    if (ctx.last_jitter() > 0) != (f.jitter > 0):
        # a pivot that is zero up to rounding: whether dpotrf (the oracle's LAPACK) or the device factorisation sees it as
        # non-positive depends on the summation order -- the two then sit on different rungs of the ladder; counted apart
        nedge += 1
        continue
    if f.jitter > 0:
        njit += 1
        check("jitter", abs(ctx.last_jitter() - f.jitter) / f.jitter, 1e-12, tag)
        bar = 1e-4                                    # a matrix at the edge of positive definiteness: conditioning eats digits on both sides
    else:
        bar = 1e-6
    check("nll", abs(nll - onll) / max(abs(onll), N / 2), bar, tag)
    check("grad", float(np.max(np.abs(g - og)) / np.max(np.abs(og))), bar if f.jitter == 0 else 1e-2, tag)
    if f.jitter == 0:
        mean, var = ctx.predict(Xs)                   # lazy refit of the factor panel at theta
        omu, ovar = go.predict(f, Xs)
        check("mean", float(np.max(np.abs(mean - omu)) / max(np.max(np.abs(omu)), 1e-300)), 1e-6, tag)
        check("var", float(np.max(np.abs(var - ovar) / np.abs(ovar))), 1e-6, tag)
    if cases % 2 == 0:                                # fit + predictions in one launch, the window in a random slot of a batch
        Mp, Bp = int(rng.integers(1, 161)), int(rng.integers(1, 6))
        slot = int(rng.integers(0, Bp))
        Xp = rng.normal(size=(Mp, d)) if kid != synth.KERNEL_RBF_BROWNIAN else (X[-1, 0] + 1.0 + np.arange(Mp, dtype=np.float64))[:, None]
        noise = bool(rng.integers(0, 2))
        XB, yB, XsB, thB = (np.stack([v] * Bp) for v in (X, y, Xp, th))
        for b in range(Bp):
            if b != slot: yB[b] = np.roll(y, b + 1)
        rc, mB, vB, lB, iB = ctxp.fit_predict_batch(XB, yB, XsB, thB, kid, include_noise=noise)
        if f.jitter == 0 and iB[slot] == 0:
            npred += 1
            omu, ovar = go.predict(f, Xp, noise)
            check("one-launch mean", float(np.max(np.abs(mB[slot] - omu)) / max(np.max(np.abs(omu)), 1e-300)), 1e-6, tag + f" M={Mp} B={Bp}")
            check("one-launch var", float(np.max(np.abs(vB[slot] - ovar) / np.abs(ovar))), 1e-6, tag + f" M={Mp} B={Bp}")
            check("one-launch logml", abs(lB[slot] - f.logml) / max(abs(f.logml), N / 2), 1e-6, tag + f" M={Mp} B={Bp}")
    if cases % 6 == 0 and f.jitter == 0 and N >= 8:
        nopt += 1
        th0 = np.ones(len(th))
        t1, l1, e1 = ctx.optimize(X, y, kid, th0, max_evals=40)
        tb, lb, eb = ctxb.optimize_batch(np.stack([X, X, X]), np.stack([y, y, y]), kid, th0, max_evals=40)
        # (a parameter the optimiser drives to exactly 0 -- the Logexp of a very negative variable: an overfitted noise variance on a
        # 13-sample, 8-dimensional window -- is equal, not 0 / 0)
        check("batch==single theta", float(np.max(np.abs(tb - t1[None]) / np.maximum(np.abs(t1[None]), 1e-300))), 1e-12, tag)
        check("batch==single logml", float(np.max(np.abs(lb - l1)) / max(abs(l1), 1.0)), 1e-12, tag)
        onl = go.nll_and_grad(kid, t1, X, y)[0]
        check("logml at optimum", abs(-l1 - onl) / max(abs(onl), N / 2), 1e-6, tag)
        check("descent", max(0.0, (-l1) - go.nll_and_grad(kid, th0, X, y)[0]) / max(abs(onl), 1.0), 1e-9, tag)
print(f"cases {cases} failures {bad} worst error / bar {worst:.3g} (optimised {nopt}, one-launch fit + predict {npred}, jitter ladder {njit}, pivot-sign edge cases skipped {nedge})")
sys.exit(1 if bad else 0)
