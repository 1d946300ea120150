#!/usr/bin/env python3
"""Where an evaluation of the one-launch short-window kernel (csrc/cgp_small.hpp) spends its cycles: per-phase s_memtime
sums of lane 0, from a -DCGP_ABLATION library (CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so).  The reference's window
(134 kept ticks, RBF x Brownian) through cgp_optimize from theta = ones."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import corenav_gp_amd.engine as engine
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "slipval_window_rbfbrownian.npz"))
t, s = g["time_array"], g["slip_array"]
n = int(0.9 * len(t))
ctx = engine.Context(max_n=256, max_m=1024, max_d=1, max_batch=1)
for _ in range(3):
    ctx.optimize(t[:n], s[:n], engine.KERNEL_RBF_BROWNIAN, np.ones(4))
t0 = time.perf_counter()
th, lml, nev = ctx.optimize(t[:n], s[:n], engine.KERNEL_RBF_BROWNIAN, np.ones(4))
el = time.perf_counter() - t0
r = ctx.debug_small()
names = ["constants", "Gram", "F (factor chain | W row, trailing)", "P (panel)", "U (next block column)", "last row of W",
         "z, alpha, logML", "gradient sums", "lane-0 step (L-BFGS)", "(evaluations)", "wave 0 chain inside F"]
ev = max(r[32 + 9], 1.0)
print(f"cgp_optimize: {1e3 * el:.3f} ms host wall, {nev} evaluations reported, {int(ev)} evaluated (jitter retries included)")
tot = 0.0
for i, nm in enumerate(names):
    print(f"  {nm:38s} {r[32 + i] / ev:9.0f} ticks per evaluation")
    tot += r[32 + i]
print(f"  {'sum':38s} {tot / ev:9.0f} ticks per evaluation  (100 MHz real-time ticks if s_memtime is the constant counter: x 10 ns)")
