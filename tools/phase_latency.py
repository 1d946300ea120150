#!/usr/bin/env python3
"""Per-phase cycle sums of the latency schedule's diagonal-tile workgroup (k_tile_sk, CGP_DBG & 1024):
   CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so CGP_DBG=1024 python tools/phase_latency.py"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
kid, X, y, Xs, th, dts = synth.config(2, batch=1)
W = bench.Workload(engine, torch, torch.device("cuda", 0), 0, kid, X, y, Xs, th, dts, 1)
for _ in range(3):
    W._call(1)
torch.cuda.synchronize(); W.ctx.debug_read()
reps = 10
for _ in range(reps):
    W._call(1)
torch.cuda.synchronize()
o = W.ctx.debug_read()
n = max(int(o[15]), 1)
names = ["update", "slab_ticket", "reduce", "lds_fill", "potf2"]
def avg(slot, cnt):
    return float(o[slot]) / max(int(o[cnt]), 1)
extra = {"panel_update_range0": avg(16, 22), "panel_update_other": avg(17, 23), "panel_last_slab_ticket": avg(18, 21),
         "panel_last_reduce": avg(19, 21), "panel_last_store": avg(20, 21), "preupdate_wg_total": avg(24, 25),
         "counts": [int(o[c]) for c in (22, 23, 21, 25)]}
print(json.dumps({"panel": extra}))
print(json.dumps({"potf2_tile_per_call": {nm: float(o[32 + i]) / max(int(o[36]), 1) for i, nm in enumerate(["F_total", "P_total", "U_total", "tail"])}, "factor_wave0_total": float(o[37]) / max(int(o[36]), 1), "helpers_wave1_total": float(o[38]) / max(int(o[36]), 1), "wave1_inverse": float(o[40]) / max(int(o[36]), 1), "wave1_trailing": float(o[41]) / max(int(o[36]), 1)}))
print(json.dumps({"trmm_sk_per_wg": {nm: float(o[44 + i]) / max(int(o[47]), 1) for i, nm in enumerate(["loads_landed", "product", "stores_issued"])}}))
print(json.dumps({"diag_wgs": n, "ticks_per_step": {nm: float(o[8 + i]) / n for i, nm in enumerate(names)}, "latency_ms": W.single_fit_latency_ms()}))
