#!/usr/bin/env python3
"""Parity at window lengths around the latency schedule's upper limit (look-ahead images cover N <= 2560) and beyond
(test infrastructure: uses oracle/): N = 2047 ... 3000, call sizes either side of the schedule switches, both
precisions, M = 130.  python tests/fuzz/check_large_n.py  (about a minute; the builder's run: 0 failures of 100 checks)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go
bad = 0
for N in (2047, 2049, 2304, 2305, 2432, 2433, 2560, 2561, 2689, 3000):
    for B, dt in ((1, engine.F64), (3, engine.F64), (17, engine.F64), (2, engine.F32), (25, engine.F32)):
        kid, X, y, Xs, th, _ = synth.config(2, batch=B, N=N, M=130)
        ctx = engine.Context(max_n=N, max_m=130, max_d=6, max_batch=B, dtype=dt)
        rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, kid)
        tol = 1e-6 if dt == engine.F64 else 1e-3
        for b in (0, B - 1):
            f = go.fit(kid, th[b], X[b], y[b]); omu, ovar = go.predict(f, Xs[b])
            e = max(abs(logml[b] - f.logml) / abs(f.logml), np.max(np.abs(mean[b] - omu)) / np.max(np.abs(omu)), np.max(np.abs(var[b] - ovar) / ovar))
            ok = rc == 0 and not info.any() and e < tol
            bad += not ok
            print("N", N, "B", B, "f64" if dt == engine.F64 else "f32", "fit", b, "err %.2e" % e, "ok" if ok else "FAIL", flush=True)
        del ctx
# round 6: dense low-dimensional fp32 windows of the same lengths -- the refined mean (two correction steps beyond 1 024 samples) at
# 2e-5, variance and logML at the fp32 bar
for N in (2049, 2561, 3000, 3073, 3500, 4200):
    for d in (1, 2, 3):
        B = 2
        Xl, yl, Xsl, thl = [], [], [], []
        for b in range(B):
            X, y, Xs = synth.window(N, d, 33, 4100 + N + b)
            Xl.append(X); yl.append(y); Xsl.append(Xs); thl.append(synth.theta_for(1, d, y, None))
        X, y, Xs, th = np.stack(Xl), np.stack(yl), np.stack(Xsl), np.stack(thl)
        ctx = engine.Context(max_n=N, max_m=33, max_d=d, max_batch=B, dtype=engine.F32)
        rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, 1)
        f = go.fit(1, th[1], X[1], y[1]); omu, ovar = go.predict(f, Xs[1])
        em = np.max(np.abs(mean[1] - omu)) / max(np.max(np.abs(omu)), 0.1 * np.max(np.abs(y[1])))
        ev = max(abs(logml[1] - f.logml) / abs(f.logml), np.max(np.abs(var[1] - ovar) / ovar))
        ok = rc == 0 and not info.any() and em < 2e-5 and ev < 1e-3
        bad += not ok
        print("N", N, "d", d, "f32 refined: mean err %.2e  var / logML err %.2e" % (em, ev), "ok" if ok else "FAIL", flush=True)
        del ctx
print("failures", bad)
