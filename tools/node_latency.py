#!/usr/bin/env python3
"""Latency of the reference node's own work item on the engine: one GP_Input window (the reference's operating size,
N = 134 ticks kept of 149, M = 599 predicted ticks, RBF x Brownian), (a) fixed theta: cgp_slip_node_callback, (b) with the
reference's m.optimize(): cgp_slip_node_callback_opt from theta = ones (gp_slip_node.py:31-49).  Host buffers in, host
buffers out, per call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import corenav_gp_amd.engine as engine
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "slipval_window_rbfbrownian.npz"))
t, s, th = g["time_array"], g["slip_array"], g["theta"]
ctx = engine.Context(max_n=256, max_m=1024, max_d=1, max_batch=1)
for _ in range(5): ctx.slip_node_callback(t, s, th)
ts = []
for _ in range(50):
    t0 = time.perf_counter(); m, sg = ctx.slip_node_callback(t, s, th); ts.append(time.perf_counter() - t0)
print(f"fixed theta: n = {len(t)}, {len(m)} predictions, median {1e6 * np.median(ts):.0f} us, min {1e6 * min(ts):.0f} us per callback")
for _ in range(2): ctx.slip_node_callback_opt(t, s, np.ones(4))
ts = []
for _ in range(10):
    t0 = time.perf_counter(); m, sg, tho = ctx.slip_node_callback_opt(t, s, np.ones(4)); ts.append(time.perf_counter() - t0)
print(f"with optimize(): median {1e3 * np.median(ts):.2f} ms, min {1e3 * min(ts):.2f} ms per callback; theta -> {np.round(tho, 5).tolist()}")
# what one evaluation of the objective costs, and how many the optimiser takes
X, y = t[:int(0.9 * len(t))], s[:int(0.9 * len(s))]
for _ in range(3): ctx.nll_grad(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
t0 = time.perf_counter()
for _ in range(100): ctx.nll_grad(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
ev = (time.perf_counter() - t0) / 100
tho, logml, nev = ctx.optimize(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
print(f"one nll + gradient evaluation (host theta in, host gradient out): {1e6 * ev:.0f} us; optimize() took {nev} evaluations")
# per-kernel time of one evaluation (the engine's own event profiler: update / diag / trsm / finalize / alpha groups)
ctx.profile_enable(True)
ctx.nll_grad(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4))
print("per-kernel time of one evaluation:", {k: (round(v["ms"], 4), v["launches"]) for k, v in ctx.profile_read().items() if v["launches"]})
ctx.profile_enable(False)
