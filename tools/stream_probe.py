#!/usr/bin/env python3
"""Which caller stream lets the two stream groups of a 64-fit fp32 call run side by side?  ms per call on ONE context with the
call enqueued on (a) the legacy default stream, (b) the context's own stream (CGP_STREAM_CTX), (c) a fresh torch stream,
each with one group (cgp_set_streams(1)) and with the engine's two groups; repeated with `--extra N` idle contexts created first."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--extra", type=int, default=0)
a = ap.parse_args()
import torch, bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
dev = torch.device("cuda", 0)
kid, X, y, Xs, th, dts = synth.config(3, batch=64)
idle = [engine.Context(max_n=256, max_m=64, max_d=1, max_batch=1) for _ in range(a.extra)]
W = bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, dts, 0)
ts = torch.cuda.Stream(dev)
def run(stream, groups, reps=100):
    W.stream = stream
    W.ctx.set_streams(groups)
    for _ in range(20): W.step()
    torch.cuda.synchronize(); W.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): W.step()
    torch.cuda.synchronize(); W.ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, st in (("legacy default stream", 0), ("context's own stream", engine.STREAM_CTX), ("a torch stream", ts.cuda_stream)):
    print(f"{name:24s} one group {run(st, 1):.4f} ms   engine's groups {run(st, 0):.4f} ms   (idle contexts before: {a.extra})", flush=True)
