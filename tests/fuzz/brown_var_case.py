#!/usr/bin/env python3
"""The sweep's pinned fp32 case of the reference's kernel (RBF x Brownian, ticks 56 ... 1079, 25 fits: variance error 3.29e-3 against
the 3e-3 bar) under the schedule the environment selects (ablation library: CGP_MID_FITS / CGP_LAT_FITS): where the error sits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go

N, M, B, kid, seed = int(os.environ.get("N", 1024)), 300, int(os.environ.get("B", 25)), 2, 250886345
VARY = os.environ.get("VARY") == "1"     # every fit of the call a different first tick (11, 13, ...): the signed errors' mean and spread
Xl, yl, Xsl, thl = [], [], [], []
for b in range(B):
    X, y, Xs = synth.window(N, 1, M, seed + b)
    X = (np.arange(N, dtype=np.float64) + 11.0 + (2 * b if VARY else seed % 50))[:, None]
    Xs = (X[-1, 0] + 1.0 + np.arange(M, dtype=np.float64))[:, None]
    Xl.append(X); yl.append(y); Xsl.append(Xs); thl.append(synth.theta_for(kid, 1, y, None))
X, y, Xs, th = np.stack(Xl), np.stack(yl), np.stack(Xsl), np.stack(thl)
ctx = engine.Context(max_n=N, max_m=M, max_d=1, max_batch=B, dtype=engine.F32)
rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, kid)
assert rc == 0 and not info.any()
worst, signed0 = 0, []
for b in range(B):
    f = go.fit(kid, th[b], X[b], y[b])
    omu, ovar = go.predict(f, Xs[b])
    ev = (var[b] - ovar) / np.abs(ovar)
    i = int(np.argmax(np.abs(ev)))
    worst = max(worst, abs(ev[i]))
    signed0.append(ev[0])
    if VARY:
        continue
    if b < 4 or abs(ev[i]) > 2e-3:
        prior = go.kernel_K(kid, th[b], Xs[b][i:i + 1])[0, 0]
        print(f"fit {b}: max |var err| {abs(ev[i]):.3e} at m={i} (signed {ev[i]:+.2e}; mean of signed errors {ev.mean():+.2e}, rms {np.sqrt((ev**2).mean()):.2e}) "
              f"prior/posterior {prior / ovar[i]:.0f} theta {th[b]}")
print(f"worst {worst:.3e}; signed error of the nearest test point over the call's fits: mean {np.mean(signed0):+.2e} sd {np.std(signed0):.2e} "
      f"min {np.min(signed0):+.2e} max {np.max(signed0):+.2e}")
