#!/usr/bin/env python3
"""Rate of the batched m.optimize() (cgp_optimize_batch) on an ensemble of the reference's windows (134 kept ticks each,
RBF x Brownian, theta from ones): windows per second and evaluations per window."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import corenav_gp_amd.engine as engine
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "slipval_window_rbfbrownian.npz"))
t, s = g["time_array"], g["slip_array"]
n = int(0.9 * len(t))
rng = np.random.default_rng(3)
for B in (1, 16, 64, 256):
    X = np.tile(t[:n, None], (B, 1, 1)).astype(np.float64)
    y = np.stack([s[:n] * (1.0 + 0.05 * rng.standard_normal()) + 0.002 * rng.standard_normal(n) for _ in range(B)])
    ctx = engine.Context(max_n=256, max_m=256, max_d=1, max_batch=B)
    ctx.optimize_batch(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
    t0 = time.perf_counter()
    th, logml, nev = ctx.optimize_batch(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
    el = time.perf_counter() - t0
    print(f"{B:4d} windows: {1e3 * el:8.2f} ms = {B / el:8.0f} windows/s, evaluations per window {np.mean(nev):.1f} (max {int(np.max(nev))})", flush=True)
