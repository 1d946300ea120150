#!/usr/bin/env python3
"""Per-phase cycle sums of k_panel (PhaseClock, CGP_DBG & 1024) from the -DCGP_ABLATION library:
   CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so CGP_DBG=1024 python tools/phase_clock.py [--config 3] [--batch 512]
prints, per block step k, the mean s_memtime ticks per workgroup of: gram (prefetch + prologue loads + Gram tile),
loop (MFMA update), fold (accumulator fold + barrier), wstage (W_k -> LDS), trmm, store."""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--batch", type=int, default=512)
a = ap.parse_args()
assert engine.load().cgp_build_flags() & engine.BUILD_ABLATION, "needs CGP_LIB=<libcorenav_gp_ab.so>"
kid, X, y, Xs, th, dts = synth.config(a.config, batch=a.batch)
W = bench.Workload(engine, torch, torch.device("cuda", 0), 0, kid, X, y, Xs, th, dts, 1)
W.step(); torch.cuda.synchronize(); W.ctx.debug_read()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); W.step(); t1.record(); torch.cuda.synchronize()
out = W.ctx.debug_read()
names = ["gram", "loop", "fold", "wstage", "trmm", "store", "gfetch"]   # gfetch: kernel entry -> Gram inputs staged; gram: the tile itself
rows = []
for k in range(32):
    s = out[64 + 8 * k: 64 + 8 * k + 8]
    if s[7] == 0:
        continue
    rows.append({"k": k, "wgs": int(s[7]), **{n: float(s[i]) / s[7] for i, n in enumerate(names)}})
print(json.dumps({"config": a.config, "batch": a.batch, "step_ms": t0.elapsed_time(t1), "ticks_per_wg": rows}))
