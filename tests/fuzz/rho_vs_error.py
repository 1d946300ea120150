#!/usr/bin/env python3
"""fp32 mean error (no refinement) against the window's density indicator rho = prior variance / geometric mean of the
pivots L_ii^2 (test infrastructure: uses oracle/): the data behind RF_RHO (csrc/cgp_kernels.hpp).
   python tests/fuzz/rho_vs_error.py [N=1100] [fits per dimension=24]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
M = 5
rows = []
for d in (3, 4, 5, 6):
    for kid in (1, 0):
        Xl, yl, Xsl, thl = [], [], [], []
        for b in range(B):
            seed = 31337 + 1000 * d + 17 * b + kid
            X, y, Xs = synth.window(N, d, M, seed)
            Xl.append(X); yl.append(y); Xsl.append(Xs)
            thl.append(synth.theta_for(kid, d, y, np.random.default_rng(seed + 7) if kid == 1 else None))
        X, y, Xs, th = np.stack(Xl), np.stack(yl), np.stack(Xsl), np.stack(thl)
        ctx = engine.Context(max_n=N, max_m=M, max_d=d, max_batch=B, dtype=engine.F32)
        ctx.set_refine(0)
        rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, kid)
        for b in range(B):
            f = go.fit(kid, th[b], X[b], y[b])
            omu, ovar = go.predict(f, Xs[b])
            ms = max(float(np.max(np.abs(omu))), 0.1 * float(np.max(np.abs(y[b]))))
            rho = (th[b][0] + th[b][-1]) * np.exp(-2.0 * np.sum(np.log(np.diag(f.L))) / N)
            rows.append((d, kid, rho, float(np.max(np.abs(mean[b] - omu))) / ms))
rows.sort(key=lambda r: r[2])
print("   d kid     rho   mean error")
for r in rows:
    print(f"{r[0]:4d} {r[1]:3d} {r[2]:7.1f}   {r[3]:.2e}")
a = np.array([(r[2], r[3]) for r in rows])
for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16), (16, 24), (24, 99)):
    s = a[(a[:, 0] >= lo) & (a[:, 0] < hi)]
    if len(s):
        print(f"rho [{lo}, {hi}): {len(s)} fits, error mean {s[:, 1].mean():.2e} max {s[:, 1].max():.2e}")
