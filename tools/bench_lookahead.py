#!/usr/bin/env python3
"""Throughput of the batched stop-time look-ahead (k_lookahead) against the host C++ loop."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from corenav_gp_amd import engine as e, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "lookahead_restated.npz"))
states = [synth.filter_state(5000 + k) for k in range(64)]
P, Q, STM, Hv, pos = (np.stack([states[k % 64][j] for k in range(T)]) for j in range(5))
means, sigmas = np.tile(g["mean"], (T, 1)), np.tile(g["sigma"], (T, 1))
P[::3] *= 1e-4   # a third of the ensemble never crosses the threshold: full 599 x 5 propagation steps
Q[::3] *= 1e-4
ctx = e.Context(max_n=8, max_m=8, max_d=1)
ctx.predict_stop_batch(means, sigmas, P, Q, STM, Hv, pos, 10.0, 10.0)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    fired, cmd, iout, xy = ctx.predict_stop_batch(means, sigmas, P, Q, STM, Hv, pos, 10.0, 10.0)
gpu = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
nh = 64
for k in range(nh):
    e.predict_stop(means[k], sigmas[k], P[k], Q[k], STM[k], Hv[k], pos[k], 10.0, 10.0)
host = (time.perf_counter() - t0) / nh
print(json.dumps({"trajectories": T, "gpu_call_ms_incl_copies": gpu * 1e3, "trajectories_per_s": T / gpu,
                  "host_cpp_ms_per_trajectory": host * 1e3, "fired": int(fired.sum()), "mean_steps": float(iout.mean())}))
