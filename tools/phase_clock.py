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
ka = out[336:344]   # kind-A workgroups (the chain tile), all block steps summed; 344.. = their diagonal-tile finish
fin = out[344:352]
extra = {}
if ka[7]:
    extra["kind_A_tile_ticks_per_wg"] = {n: float(ka[i]) / ka[7] for i, n in enumerate(names)}
    extra["kind_A_wgs"] = int(ka[7])
if fin[7]:
    extra["diag_finish_ticks_per_wg"] = {n: float(fin[i]) / fin[7] for i, n in enumerate(["fence_image_or_gram", "newest_two_block_columns", "tile_to_lds", "packed_factor_inverse_stores"])}
pf = out[32:44]      # potf2_tile (fat diagonal tile): per tile F / P / U phase sums, tail, count, wave-0 factor, helper side, inverse, trailing
if pf[4]:
    n = float(pf[4])
    extra["potf2_tile_ticks_per_tile"] = {"F_phases": pf[0] / n, "P_phases": pf[1] / n, "U_phases": pf[2] / n, "tail": pf[3] / n,
                                          "wave0_factor_in_F": pf[5] / n, "helpers_in_F": pf[6] / n, "helpers_inverse": pf[8] / n,
                                          "helpers_trailing": pf[9] / n, "wave0_block_load": pf[10] / n, "wave0_block_factor": pf[11] / n, "tiles": int(pf[4])}
print(json.dumps({"config": a.config, "batch": a.batch, "step_ms": t0.elapsed_time(t1), "ticks_per_wg": rows, **extra}))
