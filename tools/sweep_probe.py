#!/usr/bin/env python3
"""cgp_sweep_fit_predict_device over ONE device against the plain context's call, same buffers, by caller stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
dev = torch.device("cuda", 0)
kid, X, y, Xs, th, dts = synth.config(3, batch=64)
W = bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, dts, 0)
M = bench.M_TEST
def loop(fn, sync, reps=100):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3
def both():
    torch.cuda.synchronize(); W.ctx.synchronize()
print(f"context, legacy stream      {loop(W.step, both):.4f} ms")
W.stream = engine.STREAM_CTX
print(f"context, its own stream     {loop(W.step, both):.4f} ms")
sw = engine.Sweep([0], W.N, M, W.d, 64, engine.F32)
ptrs = ([W.dX.data_ptr()], [W.dy.data_ptr()], [W.dXs.data_ptr()], [W.dth.data_ptr()], None, True,
        [W.dmean.data_ptr()], [W.dvar.data_ptr()], [W.dlogml.data_ptr()], [W.dinfo.data_ptr()])
def swsync():
    sw.synchronize(); torch.cuda.synchronize()
for name, st in (("contexts' own streams", None), ("legacy stream", [0]), ("a torch stream", [torch.cuda.Stream(dev).cuda_stream])):
    print(f"sweep [0], {name:22s} {loop(lambda: sw.fit_predict_device(64, W.N, W.d, M, kid, *ptrs, st), swsync):.4f} ms")
t0 = time.perf_counter()
for _ in range(200): sw.fit_predict_device(64, W.N, W.d, M, kid, *ptrs, None)
host = (time.perf_counter() - t0) / 200 * 1e3
swsync()
print(f"host time to issue one sweep call: {host:.4f} ms")
sw.set_streams(1)
print(f"sweep [0], one group           {loop(lambda: sw.fit_predict_device(64, W.N, W.d, M, kid, *ptrs, None), swsync):.4f} ms")
