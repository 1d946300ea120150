#!/usr/bin/env python3
"""How long does a k_panel launch really take?  Three clocks on the same workload (ablation build):
  a) in-kernel real-time stamps (CGP_DBG=2048: earliest workgroup entry .. latest exit per block step), no events;
  b) HIP events around every launch (cgp_profile_enable(1)) -- what bench.py's roofline used in round 1;
  c) HIP events around ONE launch per step, the others back to back (cgp_profile_enable(2 + k)).
   CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so CGP_DBG=2048 python tools/launch_spans.py [--config 3] [--batch 512]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
NT = (X.shape[1] + 127) // 128
for _ in range(3):
    W.step()
torch.cuda.synchronize(); W.ctx.debug_read()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); W.step(); t1.record(); torch.cuda.synchronize()
out = W.ctx.debug_read()
spans = [(int(out[416 + k]) - ((1 << 62) - int(out[384 + k]))) * 1e-5 for k in range(NT)]   # 100 MHz ticks -> ms
starts = [((1 << 62) - int(out[384 + k])) for k in range(NT)]
gaps = [(starts[k + 1] - int(out[416 + k])) * 1e-5 for k in range(NT - 1)]
wg = [int(out[64 + 8 * k + 7]) for k in range(NT)]          # needs CGP_DBG & 1024 too (PhaseClock's workgroup count)
resid = [int(out[448 + k]) * 1e-5 for k in range(NT)]        # sum over workgroups of their residence, ms
clock = [int(out[480 + k]) / max(int(out[448 + k]), 1) / 10.0 for k in range(NT)]   # s_memtime ticks per ns
W.ctx.profile_enable(True)
W.step(); torch.cuda.synchronize()
allev = W.ctx.profile_read()["update"]
one = []
for k in range(NT):
    W.ctx.lib.cgp_profile_enable(W.ctx.h, 2 + k)
    W.step(); torch.cuda.synchronize()
    u = W.ctx.profile_read()["update"]
    one.append(u["ms"])
W.ctx.profile_enable(False)
print(json.dumps({"config": a.config, "batch": a.batch, "step_ms": t0.elapsed_time(t1),
                  "in_kernel_span_ms": [round(x, 4) for x in spans], "in_kernel_sum_ms": sum(spans),
                  "gaps_between_launches_ms": [round(x, 4) for x in gaps],
                  "workgroups_counted": wg, "mean_residence_ms": [round(r / max(w, 1), 4) for r, w in zip(resid, wg)],
                  "mean_resident_workgroups_per_cu": [round(r / max(sp, 1e-9) / 256, 3) for r, sp in zip(resid, spans)],
                  "memtime_ticks_per_ns": [round(c, 3) for c in clock],
                  "events_every_launch_sum_ms": allev["ms"], "events_one_launch_per_step_ms": [round(x, 4) for x in one],
                  "events_one_launch_sum_ms": sum(one)}))
