#!/usr/bin/env python3
"""Streaming benchmark of the sliding-window GP (BASELINE configs[3]: N = 512 ring, rank-1 update
per tick).  Bound: HBM/L2 traffic -- the factor is read and written once per tick
(algorithmic bytes per tick = n^2/2 * 8 B * 2 = 2.1 MB at n = 512)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--d", type=int, default=3)
ap.add_argument("--windows", type=int, default=256)
ap.add_argument("--ticks", type=int, default=2000)
args = ap.parse_args()
import torch
import corenav_gp_amd.engine as engine
dev = torch.device("cuda", 0)
W, N, d, T = args.windows, args.n, args.d, args.ticks
rng = np.random.default_rng(20264)
t = np.arange(11, 11 + N + T, dtype=np.float64)
X = np.stack([np.column_stack([(t - t.mean()) / t.std()] + [rng.normal(size=len(t)) for _ in range(d - 1)]) for _ in range(W)])
y = 0.1 * np.sin(2 * np.pi * t / 40.0)[None] + rng.normal(0, 0.03, (W, len(t)))
theta = np.concatenate([[0.02], np.linspace(0.8, 1.6, d), [1e-3]])
ctx = engine.Context(max_n=8, max_m=8, max_d=d)
ctx.window_init(W, N, d, 1, theta)
dX, dy = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
def push(a, b):
    xs = dX[:, a:b].contiguous(); ys = dy[:, a:b].contiguous()
    out = torch.empty((3, W, b - a), device=dev, dtype=torch.float64)
    ctx.window_push_device(b - a, xs.data_ptr(), ys.data_ptr(), True, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    return out
push(0, N)                       # fill the windows (warm-up, not timed)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = push(N, N + T)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bytes_tick = N * N / 2 * 8 * 2
print(json.dumps({"metric": "window-ticks/s", "value": W * T / dt, "windows": W, "N": N, "d": d, "ticks": T,
                  "us_per_tick_per_window": dt / T * 1e6, "algorithmic_GBps": W * T * bytes_tick / dt / 1e9,
                  "hbm_peak_GBps": 8000, "frac": W * T * bytes_tick / dt / 8e12, "info": ctx.window_state(0)[1],
                  "logml_last": float(out[2, 0, -1])}))
