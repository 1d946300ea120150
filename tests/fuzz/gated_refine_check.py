#!/usr/bin/env python3
"""The per-fit density gate of the fp32 refinement at d = 4 (test infrastructure: uses oracle/): the windows round 6's sweep flagged
(rho = 21) inside a batch with sparser ones -- marked fits are refined (1e-7), unmarked fits are bitwise what cgp_set_refine(0) gives.
   python tests/fuzz/gated_refine_check.py"""
import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go
# d = 4 windows: the two the sweep flagged (rho = 21) inside a batch of sparse ones
N, M, d, B = 1100, 5, 4, 25
Xl, yl, Xsl, thl = [], [], [], []
seed = 642426859
for b in range(B):
    X, y, Xs = synth.window(N, d, M, seed + b)
    Xl.append(X); yl.append(y); Xsl.append(Xs); thl.append(synth.theta_for(1, d, y, np.random.default_rng(seed + 7 + b)))
X, y, Xs, th = np.stack(Xl), np.stack(yl), np.stack(Xsl), np.stack(thl)
ctx = engine.Context(max_n=N, max_m=M, max_d=d, max_batch=B, dtype=engine.F32)
out = {}
for mode in (-1, 0, 1):
    ctx.set_refine(mode)
    rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, 1)
    assert rc == 0
    out[mode] = mean
errs = {m: [] for m in out}
rhos = []
for b in range(B):
    f = go.fit(1, th[b], X[b], y[b]); omu, _ = go.predict(f, Xs[b])
    ms = max(float(np.max(np.abs(omu))), 0.1 * float(np.max(np.abs(y[b]))))
    rhos.append((th[b][0] + th[b][-1]) * np.exp(-2 * np.sum(np.log(np.diag(f.L))) / N))
    for m in out: errs[m].append(float(np.max(np.abs(out[m][b] - omu))) / ms)
for b in range(B):
    print(f"fit {b:2d} rho {rhos[b]:5.1f}  err default {errs[-1][b]:.2e}  off {errs[0][b]:.2e}  forced {errs[1][b]:.2e}", "MARKED" if rhos[b] >= 12 else "")
    if rhos[b] >= 12.5: assert errs[-1][b] < 2e-5, b
    if rhos[b] < 11.5: assert errs[-1][b] == errs[0][b], b
print("ok")
