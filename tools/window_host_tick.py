import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
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
for i in range(20): c1.window_push(X[:1, N + i:N + i + 1], y[:1, N + i:N + i + 1])
best = 1e9
for rep in range(3):
    t1 = time.perf_counter()
    for i in range(20 + 80 * rep, 20 + 80 * rep + 80): c1.window_push(X[:1, N + i:N + i + 1], y[:1, N + i:N + i + 1])
    best = min(best, (time.perf_counter() - t1) / 80 * 1e6)
print(f"host tick {best:.1f} us (one N = 512 window, T = 1 pushes)")
