#!/usr/bin/env python3
"""Host time to ISSUE one call against the device time it takes (fp32 N = 1024, 64 fits; fp64 N = 2048, 64 fits): is a
mid-size call launch-bound?  usage: python3 tools/host_issue_time.py [streams=2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
import corenav_gp_amd.engine as engine
import corenav_gp_amd.synth as synth
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
for cfg, N, B in ((3, 1024, 64), (2, 2048, 64), (3, 1024, 512)):
    kid, X, y, Xs, th, dts = synth.config(cfg, batch=B, N=N, M=bench.M_TEST)
    w = bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, dts, streams)
    for _ in range(20): w.step()
    torch.cuda.synchronize()
    # device-bound rate: many calls queued
    t0 = time.perf_counter()
    for _ in range(100): w.step()
    t_issue_all = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # host issue cost with an idle device in front: issue one call, sync, repeat
    iss = []
    for _ in range(30):
        t0 = time.perf_counter(); w.step(); iss.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print(f"{dts} N={N} fits={B} streams={streams}: {1e3 * t_all / 100:.3f} ms per call queued back to back (host spent {1e3 * t_issue_all / 100:.3f} ms per call issuing, "
          f"queue full = it waits); issue alone with an idle device {1e3 * np.median(iss):.3f} ms")
