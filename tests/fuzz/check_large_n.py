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
print("failures", bad)
