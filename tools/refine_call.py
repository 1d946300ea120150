#!/usr/bin/env python3
"""One refined fp32 call in a loop (for rocprofv3): python tools/refine_call.py [--n 1024] [--d 1] [--batch 512] [--reps 10]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from corenav_gp_amd import engine, synth
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024); ap.add_argument("--d", type=int, default=1)
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--reps", type=int, default=10); ap.add_argument("--m", type=int, default=599)
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, M, d, B = a.n, a.m, a.d, a.batch
Xl, yl, Xsl, thl = [], [], [], []
for b in range(B):
    X, y, Xs = synth.window(N, d, M, 777 + b)
    th = np.zeros(engine.MAX_THETA); t = synth.theta_for(1, d, y, None); th[:len(t)] = t
    Xl.append(X.T.copy()); yl.append(y); Xsl.append(Xs.T.copy()); thl.append(th)
f32 = lambda v: torch.tensor(np.stack(v), dtype=torch.float32, device=dev)
dX, dy, dXs = f32(Xl), f32(yl), f32(Xsl)
dth = torch.tensor(np.stack(thl), dtype=torch.float64, device=dev)
dmean, dvar = torch.zeros(B, M, dtype=torch.float32, device=dev), torch.zeros(B, M, dtype=torch.float32, device=dev)
dl, di = torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
ctx = engine.Context(max_n=N, max_m=M, max_d=d, max_batch=B, dtype=engine.F32)
for _ in range(a.reps):
    ctx.fit_predict_batch_device(B, N, d, M, 1, dX.data_ptr(), dy.data_ptr(), dXs.data_ptr(), dth.data_ptr(), 0, 1, dmean.data_ptr(), dvar.data_ptr(), dl.data_ptr(), di.data_ptr(), 0)
torch.cuda.synchronize()
assert int(di.abs().sum()) == 0
print("ok")
