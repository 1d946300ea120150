#!/usr/bin/env python3
"""Do two mid-size calls on two HIP streams (two engine contexts, two sets of slabs) overlap on the GPU?
   python tools/pipeline_probe.py [--config 3] [--batch 64] [--depth 2]
Prints ms per call for depth 1 (calls back to back on one stream) and depth D (call i on stream i % D)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--calls", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda", 0)
kid, X, y, Xs, th, dts = synth.config(a.config, batch=a.batch)
out = {"config": a.config, "batch": a.batch}
for depth in (1, a.depth):
    Ws = [bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, dts, 1) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for W, s in zip(Ws, streams):
        W.stream = s.cuda_stream
    def run(n):
        for i in range(n):
            Ws[i % depth].step()
    run(2 * depth); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(a.calls); torch.cuda.synchronize()
    out[f"ms_per_call_depth{depth}"] = (time.perf_counter() - t0) / a.calls * 1e3
    del Ws
print(json.dumps(out))
