#!/usr/bin/env python3
"""Device-memory leak check: 150 contexts (alternating precision; batch call + sliding windows each) and 20 two-context
sweeps created and destroyed; free device memory must not drift (builder's run: constant after the first context)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from corenav_gp_amd import engine, synth
kid, X, y, Xs, th, _ = synth.config(2, batch=4, N=300)
free0 = torch.cuda.mem_get_info()[0]
for i in range(150):
    ctx = engine.Context(max_n=512, max_m=599, max_d=6, max_batch=32, dtype=engine.F64 if i % 2 else engine.F32)
    ctx.fit_predict_batch(X, y, Xs, th, kid)
    ctx.window_init(2, 40, 6, kid, th[0]); 
    ctx.window_push(np.random.randn(2, 10, 6), np.random.randn(2, 10))
    del ctx
    if i % 50 == 49:
        print(i, "free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6, flush=True)
sw = [engine.Sweep([0, 0], 300, 599, 6, 8) for _ in range(20)]
del sw
print("after sweeps free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6)
