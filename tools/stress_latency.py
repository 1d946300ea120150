#!/usr/bin/env python3
"""Stress test of the split-K latency schedule: the ticketed reduction must give bitwise identical
results on every run (partials are added in range order), for 1..24 fits per call, fp64 and fp32."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from corenav_gp_amd import engine, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
# short windows keep the schedule up to larger calls (32 fits at one or two block steps, 28 / 22 at four / six in fp64)
for cfg, dtype, N in ((2, engine.F64, 2048), (2, engine.F64, 1000), (3, engine.F32, 1024), (2, engine.F64, 134), (2, engine.F64, 500),
                      (2, engine.F64, 700), (3, engine.F32, 200), (3, engine.F32, 512)):
    for B in ((1, 3, 4, 13, 24) if N > 800 else (1, 5, 22, 28, 32)):
        kid, X, y, Xs, th, _ = synth.config(cfg, batch=B, N=N)
        ctx = engine.Context(max_n=N, max_m=Xs.shape[1], max_d=X.shape[2], max_batch=B, dtype=dtype)
        ref = ctx.fit_predict_batch(X, y, Xs, th, kid)
        for i in range(reps):
            r = ctx.fit_predict_batch(X, y, Xs, th, kid)
            if not (np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2]) and np.array_equal(r[3], ref[3])):
                bad += 1
                print("MISMATCH cfg", cfg, "N", N, "B", B, "rep", i, float(np.abs(r[1] - ref[1]).max()))
        print("cfg", cfg, "N", N, "B", B, "ok" if not bad else "BAD", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
