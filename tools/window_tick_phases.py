#!/usr/bin/env python3
"""Where one host tick of one N = 512 window goes inside `k_window_ticks`: needs a `make variant NAME=winp EXTRA=-DWIN_PROBE=1` library
(CGP_LIB=corenav_gp_amd/libcorenav_gp_winp.so) whose tick outputs are wave 0's s_memtime sums: wait at the panel barrier, its 32 loads
landed, the 16 rows' sweep + stores, the serial diagonal block (phase A); 32 panels per tick."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch  # noqa: F401
from corenav_gp_amd import engine
N, d = 512, 3
rng = np.random.default_rng(20264)
T = N + 300
t = np.arange(11, 11 + T, dtype=np.float64)
X = np.empty((1, T, d)); X[:, :, 0] = (t - t.mean()) / t.std(); X[:, :, 1:] = rng.normal(size=(1, T, d - 1))
y = 0.1 * np.sin(2 * np.pi * t / 40.0)[None] + rng.normal(0, 0.03, (1, T))
theta = np.concatenate([[0.02], np.linspace(0.8, 1.6, d), [1e-3]])
c1 = engine.Context(max_n=8, max_m=8, max_d=d)
c1.window_init(1, N, d, 1, theta)
c1.window_push(X[:1, :N], y[:1, :N])
acc = np.zeros(5); n = 0
t0 = time.perf_counter()
for i in range(200):
    pm, pv, lm = c1.window_push(X[:1, N + i:N + i + 1], y[:1, N + i:N + i + 1])
    if i >= 20:
        a, b = divmod(lm[0, 0], 67108864.0); c, e = divmod(pm[0, 0], 67108864.0)
        acc += [a, b, c, e, pv[0, 0]]; n += 1
host = (time.perf_counter() - t0) / 200 * 1e6
acc /= n
print(f"host tick {host:.1f} us; wave 0 per tick, shader cycles (s_memtime, 2.4 GHz): barrier wait {acc[0]:.0f}, loads landed {acc[1]:.0f}, row sweep + stores {acc[2]:.0f}, "
      f"serial block {acc[3]:.0f}; panel loop {acc[4]:.0f} = {acc[4] / 2400:.1f} us")
