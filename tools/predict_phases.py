#!/usr/bin/env python3
"""Where the one-launch fit + predict kernel of a short window (csrc/cgp_small.hpp: k_small_predict) spends its cycles: per-phase
s_memtime sums of workgroup 0 / lane 0 from a -DCGP_ABLATION library (CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so), for the
reference's own callback (149-tick GP_Input, 599 predictions, fixed theta); `batch`: 256 such windows in one call (window 0's clocks)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import corenav_gp_amd.engine as engine
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "slipval_window_rbfbrownian.npz"))
t, s, th = g["time_array"], g["slip_array"], g["theta"]
batch = len(sys.argv) > 1 and sys.argv[1] == "batch"   # 256 windows: one workgroup per window, all 38 chunks in its loop
ctx = engine.Context(max_n=256, max_m=1024, max_d=1, max_batch=256 if batch else 1)
for _ in range(5): ctx.slip_node_callback(t, s, th)
ts = []
for _ in range(50):
    t0 = time.perf_counter(); ctx.slip_node_callback(t, s, th); ts.append(time.perf_counter() - t0)
print(f"callback: median {1e6 * np.median(ts):.0f} us")
if batch:
    n, W = int(0.9 * len(t)), 256
    X = np.stack([t[:n] + k for k in range(W)])[:, :, None]
    y = np.stack([np.roll(s, k)[:n] for k in range(W)])
    Xs = np.stack([X[k, -1, 0] + 1 + np.arange(599.0) for k in range(W)])[:, :, None]
    for _ in range(3): ctx.fit_predict_batch(X, y, Xs, np.tile(th, (W, 1)), engine.KERNEL_RBF_BROWNIAN)
    tb = []
    for _ in range(10):
        t0 = time.perf_counter(); ctx.fit_predict_batch(X, y, Xs, np.tile(th, (W, 1)), engine.KERNEL_RBF_BROWNIAN); tb.append(time.perf_counter() - t0)
    print(f"{W} windows in one call, host buffers: median {1e3 * np.median(tb):.3f} ms")
r = ctx.debug_small()   # batch: window 0's record of the batched launch
names = {0: "constants", 1: "Gram", 2: "F (factor chain | W row, trailing)", 3: "P (panel)", 5: "last row of W", 6: "z, alpha, logML",
         10: "staging (window, work lists, theta, constants)", 11: "K* chunks", 12: "means, V = W K*, |V|^2", 13: "outputs"}
tot = 0.0
for i, nm in names.items():
    print(f"  {nm:42s} {r[32 + i]:9.0f} ticks")
    tot += r[32 + i]
print(f"  {'sum':42s} {tot:9.0f} ticks (x 10 ns if s_memtime is the 100 MHz counter)")
