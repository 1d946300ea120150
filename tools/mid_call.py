#!/usr/bin/env python3
"""One mid-size call in a loop, for timelines and A/Bs:  python tools/mid_call.py [--config 3] [--batch 64] [--streams 1] [--reps 200]
prints ms per call (torch events around `reps` back-to-back calls on ONE context).  Under `rocprofv3 --kernel-trace` the
trace holds the calls' launches (tools/mid_timeline.sh condenses the last call into a table)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--streams", type=int, default=0, help="cgp_set_streams (0 = leave the default)")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--n", type=int, default=None)
ap.add_argument("--refine", type=int, default=None, help="cgp_set_refine (fp32: correction steps of alpha / the mean)")
ap.add_argument("--sweep", action="store_true", help="the same call through cgp_sweep_fit_predict_device over [0] (a second context), its own streams")
ap.add_argument("--torch-stream", action="store_true", help="enqueue on a fresh torch stream instead of the legacy default stream")
a = ap.parse_args()
import torch, bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
kw = {} if a.n is None else {"N": a.n}
kid, X, y, Xs, th, dts = synth.config(a.config, batch=a.batch, **kw)
W = bench.Workload(engine, torch, torch.device("cuda", 0), 0, kid, X, y, Xs, th, dts, a.streams)
if a.streams:
    W.ctx.set_streams(a.streams)
if a.refine is not None:
    W.ctx.set_refine(a.refine)
if a.torch_stream:
    ts = torch.cuda.Stream(torch.device("cuda", 0))
    W.stream = ts.cuda_stream
step = W.step
if a.sweep:
    sw = engine.Sweep([0], W.N, bench.M_TEST, W.d, a.batch, engine.F32 if dts == "f32" else engine.F64)
    ptrs = ([W.dX.data_ptr()], [W.dy.data_ptr()], [W.dXs.data_ptr()], [W.dth.data_ptr()], None, True,
            [W.dmean.data_ptr()], [W.dvar.data_ptr()], [W.dlogml.data_ptr()], [W.dinfo.data_ptr()])
    step = lambda: sw.fit_predict_device(a.batch, W.N, W.d, bench.M_TEST, kid, *ptrs, None)
class _S:
    pass
W2 = _S(); W2.step = step
for _ in range(20): W2.step()
torch.cuda.synchronize()
if a.sweep: sw.synchronize()
best = 1e9
for _ in range(3):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.reps): W2.step()
    if a.sweep: sw.synchronize()
    t1.record(); torch.cuda.synchronize()
    best = min(best, t0.elapsed_time(t1) / a.reps)
print(f"{best:.4f} ms per call (batch {a.batch}, config {a.config}, streams {a.streams or 'default'})")
