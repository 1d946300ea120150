#!/usr/bin/env python3
"""Latency of the reference's real operating point on the HIP path: one 149-tick window (134
training samples, RBF x Brownian, 599 predictions), fixed theta and with the optimiser."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corenav_gp_amd.engine as engine, corenav_gp_amd.synth as synth
t, s = synth.reference_window()
ctx = engine.Context(max_n=256, max_m=1024, max_d=1)
theta = np.array([0.5, 30.0, 0.01, 0.002])
for _ in range(3):
    ctx.slip_node_callback(t, s, theta)
t0 = time.perf_counter(); n = 50
for _ in range(n):
    ctx.slip_node_callback(t, s, theta)
fixed = (time.perf_counter() - t0) / n
ctx.slip_node_callback_opt(t, s, np.ones(4))
t0 = time.perf_counter(); n2 = 5
for _ in range(n2):
    m, sg, th = ctx.slip_node_callback_opt(t, s, np.ones(4))
opt = (time.perf_counter() - t0) / n2
X = t[:134, None]
th2, lml, nev = ctx.optimize(X, s[:134], 2, np.ones(4))
B = 256
Xb = np.stack([synth.reference_window(149, tick0=11 + b, seed=synth.SEED_BASE + b)[0][:134, None] for b in range(B)])
yb = np.stack([synth.reference_window(149, tick0=11 + b, seed=synth.SEED_BASE + b)[1][:134] for b in range(B)])
ctxb = engine.Context(max_n=134, max_m=134, max_d=1, max_batch=B)
ctxb.optimize_batch(Xb[:8], yb[:8], 2, np.ones(4))
t0 = time.perf_counter()
thb, lmlb, nevb = ctxb.optimize_batch(Xb, yb, 2, np.ones(4))
tb = time.perf_counter() - t0
print(json.dumps({"batched_optimiser": {"windows": B, "wall_ms": tb * 1e3, "windows_per_s": B / tb, "max_evals": int(nevb.max()),
                                        "mean_evals": float(nevb.mean())}}))
print(json.dumps({"window_ticks": 149, "n_train": 134, "m_pred": 599, "callback_fixed_theta_ms": fixed * 1e3,
                  "callback_with_optimiser_ms": opt * 1e3, "optimiser_evals": nev, "theta_opt": th2.tolist(), "logml": lml}))
