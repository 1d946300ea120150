#!/usr/bin/env python3
"""Timing probes of a mid-size call (ablation build; results are wrong on purpose):
   CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so python tools/probe_mid.py [--config 3] [--batch 64]
one line per CGP_DBG probe: ms per call."""
import argparse, json, os, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--child", type=int, default=-1)
a = ap.parse_args()
if a.child < 0:
    probes = [(0, "none"), (65536, "no diagonal-tile work inside the launches"), (64, "no in-register trmm"), (16384, "no Gram tile"),
              (32768, "no tile store"), (4096, "row panels from an L2-resident slab")]
    for dbg, name in probes:
        env = dict(os.environ, CGP_DBG=str(dbg))
        out = subprocess.run([sys.executable, __file__, "--config", str(a.config), "--batch", str(a.batch), "--child", str(dbg)],
                             env=env, capture_output=True, text=True)
        print(f"{name:48s} {out.stdout.strip()}", flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
assert engine.load().cgp_build_flags() & engine.BUILD_ABLATION, "needs CGP_LIB=<libcorenav_gp_ab.so>"
kid, X, y, Xs, th, dts = synth.config(a.config, batch=a.batch)
W = bench.Workload(engine, torch, torch.device("cuda", 0), 0, kid, X, y, Xs, th, dts, 1)
for _ in range(20): W.step()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(100): W.step()
t1.record(); torch.cuda.synchronize()
print(f"{t0.elapsed_time(t1) / 100:.4f} ms per call")
