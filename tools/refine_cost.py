#!/usr/bin/env python3
"""Cost of the fp32 refinement (cgp_set_refine) per device-resident call: ms per call without / with 1 / with 2 steps, for
call sizes and input dimensions:  python tools/refine_cost.py [--n 1024] [--m 599]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from corenav_gp_amd import engine, synth
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--m", type=int, default=599)
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
for d, kid in ((1, 1), (2, 1), (3, 1), (6, 1), (1, 2)):
    for B in (64, 512):
        N, M = a.n, a.m
        Xl, yl, Xsl, thl = [], [], [], []
        for b in range(B):
            X, y, Xs = synth.window(N, d, M, 777 + b)
            if kid == 2:
                X = (np.arange(N, dtype=np.float64) + 11.0)[:, None]
                Xs = (X[-1, 0] + 1.0 + np.arange(M, dtype=np.float64))[:, None]
            th = np.zeros(engine.MAX_THETA); t = synth.theta_for(kid, d, y, None); th[:len(t)] = t
            Xl.append(X.T.copy()); yl.append(y); Xsl.append(Xs.T.copy()); thl.append(th)
        f32 = lambda v: torch.tensor(np.stack(v), dtype=torch.float32, device=dev)
        dX, dy, dXs = f32(Xl), f32(yl), f32(Xsl)
        dth = torch.tensor(np.stack(thl), dtype=torch.float64, device=dev)
        dmean, dvar = torch.zeros(B, M, dtype=torch.float32, device=dev), torch.zeros(B, M, dtype=torch.float32, device=dev)
        dl, di = torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        ctx = engine.Context(max_n=N, max_m=M, max_d=d, max_batch=B, dtype=engine.F32)
        out = []
        for steps in (0, 1, 2):
            ctx.set_refine(steps)
            call = lambda: ctx.fit_predict_batch_device(B, N, d, M, kid, dX.data_ptr(), dy.data_ptr(), dXs.data_ptr(), dth.data_ptr(), 0, 1,
                                                        dmean.data_ptr(), dvar.data_ptr(), dl.data_ptr(), di.data_ptr(), 0)
            for _ in range(12): call()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps): call()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.reps)
            out.append(best)
        assert int(di.abs().sum()) == 0
        print(f"N={N} M={M} d={d} kid={kid} fits={B}: ms per call  refine 0 / 1 / 2 steps  {out[0]:.3f} / {out[1]:.3f} / {out[2]:.3f}   (+{100*(out[1]/out[0]-1):.0f} % / +{100*(out[2]/out[0]-1):.0f} %)")
        ctx.close()
