#!/usr/bin/env python3
"""The "slow context" of round 3: the same 64-fit fp64 call ran at 8.4 instead of 6.8 ms on a
context created right after a 512-fit context.  Every scenario below runs in a FRESH process and prints the ms per
64-fit call of the context under test next to the device addresses of its buffers:

    fresh            the 64-fit context alone
    after512         a 512-fit context created and stepped first (the round-3 observation)
    after512_s<K>    the same, with K throw-away HIP streams created between the two contexts (K = 1, 2, 3): shifts
                     the HIP-stream -> hardware-queue assignment of the second context's worker streams, not its memory
    after512_free    the 512-fit context destroyed before the 64-fit one is created (memory placement without the
                     first context's streams)
    before512        the 64-fit context created FIRST, then the 512-fit one, then timed
    *_ctxstream      the call issued on the context's private stream instead of the legacy default stream

usage: python3 tools/ctx_placement.py [--dtype f64|f32] [scenario ...]      (no scenario: all of them, one child each)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENARIOS = ["fresh", "after512", "after512_s1", "after512_s2", "after512_s3", "after512_free", "before512",
             "fresh_ctxstream", "after512_ctxstream"]


def child(scn, dtype):
    import torch
    import bench
    import corenav_gp_amd.engine as engine
    import corenav_gp_amd.synth as synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = 2 if dtype == "f64" else 3
    kid, X, y, Xs, th, dts = synth.config(cfg, batch=512, M=bench.M_TEST)
    mk = lambda n: bench.Workload(engine, torch, dev, 0, kid, X[:n], y[:n], Xs[:n], th[:n], dts, 1)
    base = scn.replace("_ctxstream", "")
    big = None
    if base == "before512":
        w = mk(64)
        big = mk(512)
    elif base == "fresh":
        w = mk(64)
    else:
        big = mk(512)
    if big is not None:
        for _ in range(3):
            big.step()
        torch.cuda.synchronize()
    if base.startswith("after512_s"):
        hip = ctypes.CDLL("libamdhip64.so")
        keep = []
        for _ in range(int(base[-1])):
            s = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
            keep.append(s)
    if base == "after512_free":
        big.ctx.close()
        big = None
        torch.cuda.synchronize()
    if base not in ("before512", "fresh"):
        w = mk(64)
    if scn.endswith("_ctxstream"):
        w.stream = engine.STREAM_CTX
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        w.step()
        w.ctx.synchronize()
        torch.cuda.synchronize()
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        w.step()
    w.ctx.synchronize()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    assert int(w.dinfo.abs().sum().item()) == 0
    bufs = w.ctx.debug_buffers()
    print(json.dumps({"scenario": scn, "dtype": dtype, "ms_per_64_fit_call": round(ms, 4),
                      "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES"),
                      "buffers": {n: [hex(a), b] for n, (a, b) in zip(
                          ("Lw", "Winv", "diagimg", "panimg", "X", "macc", "latpart", "latimg"), bufs)}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--child", default=None)
    ap.add_argument("scenarios", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.dtype)
        return
    for scn in (a.scenarios or SCENARIOS):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", a.dtype, "--child", scn],
                           capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(lines[-1] if lines else json.dumps({"scenario": scn, "error": (r.stderr or r.stdout)[-400:]}), flush=True)


if __name__ == "__main__":
    main()
