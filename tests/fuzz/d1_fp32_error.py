#!/usr/bin/env python3
"""Error of fp32 fits of dense one-dimensional windows against the fp64 oracle, all fits of a few calls (test infrastructure: uses oracle/):
   CGP_LIB=<lib> python tests/fuzz/d1_fp32_error.py      -- same-box A/B of library builds (e.g. `make variant` builds) on the windows the
fuzz sweep flagged (tests/fuzz/fuzz_parity.py's generator: seeds below)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine, synth
from oracle import gp_oracle as go

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default=None, help='"N,M,d,kid,B,seed;..." instead of the flagged windows')
ap.add_argument("--f64", action="store_true", help="an fp64 context (sanity: 1e-9)")
ap.add_argument("--refine", type=int, default=None, help="cgp_set_refine (default: the engine's own choice)")
ap.add_argument("--every", type=int, default=3, help="compare every n-th fit with the oracle")
ap.add_argument("--noise", type=float, default=None, help="override the noise variance of every window (the generator's is 3e-4 at an amplitude of 1e-2)")
a = ap.parse_args()
CASES = [(1000, 1, 1, 1, 40, 945332210), (1000, 1, 1, 1, 40, 472841713), (700, 1, 1, 1, 48, 384559499), (1024, 1, 2, 1, 38, 986455068),
         (1100, 599, 1, 1, 33, 681215296), (1000, 5, 1, 1, 40, 12345)]
if a.cases:
    CASES = [tuple(int(v) for v in c.split(",")) for c in a.cases.split(";")]
for N, M, d, kid, B, seed in CASES:
    Xl, yl, Xsl, thl = [], [], [], []
    for b in range(B):
        X, y, Xs = synth.window(N, d, max(M, 1), seed + b)
        if kid == synth.KERNEL_RBF_BROWNIAN:     # the fuzz sweep's inputs for the reference's kernel: raw tick counts
            X = (np.arange(N, dtype=np.float64) + 11.0 + (seed % 50))[:, None]
            Xs = (X[-1, 0] + 1.0 + np.arange(max(M, 1), dtype=np.float64))[:, None]
        Xl.append(X); yl.append(y); Xsl.append(Xs[:M]); thl.append(synth.theta_for(kid, d, y, None))
        if a.noise is not None:
            thl[-1][-1] = a.noise
    X, y, th, Xs = np.stack(Xl), np.stack(yl), np.stack(thl), np.stack(Xsl)
    ctx = engine.Context(max_n=N, max_m=M, max_d=d, max_batch=B, dtype=engine.F64 if a.f64 else engine.F32)
    if a.refine is not None:
        ctx.set_refine(a.refine)
    rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, kid)
    errs = []
    for b in range(0, B, a.every):
        f = go.fit(kid, th[b], X[b], y[b])
        omu, ovar = go.predict(f, Xs[b])
        mscale = max(float(np.max(np.abs(omu))), 0.1 * float(np.max(np.abs(y[b]))))
        errs.append((abs(logml[b] - f.logml) / max(abs(f.logml), 0.5 * N), float(np.max(np.abs(mean[b] - omu)) / mscale),
                     float(np.max(np.abs(var[b] - ovar) / np.abs(ovar)))))
    comp = np.array(errs)
    errs = comp.max(axis=1)
    print(f"N={N} M={M} d={d} kid={kid} B={B} seed={seed}: error over {len(errs)} fits  mean {errs.mean():.3e}  median {np.median(errs):.3e}  max {errs.max():.3e}"
          f"   by output (mean / max over fits): logML {comp[:, 0].mean():.2e} / {comp[:, 0].max():.2e}  mean {comp[:, 1].mean():.2e} / {comp[:, 1].max():.2e}"
          f"  variance {comp[:, 2].mean():.2e} / {comp[:, 2].max():.2e}")
