#!/usr/bin/env python3
"""One-fit calls in a loop for rocprofv3 --kernel-trace (per-step kernel durations of the latency schedule)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
kid, X, y, Xs, th, dts = synth.config(int(sys.argv[1]) if len(sys.argv) > 1 else 2, batch=1)
W = bench.Workload(engine, torch, torch.device("cuda", 0), 0, kid, X, y, Xs, th, dts, 1)
for _ in range(6):
    W._call(1)
    torch.cuda.synchronize()
print("latency_ms", W.single_fit_latency_ms())
