#!/usr/bin/env python3
"""Where the host-buffer batch call spends its time (ablation build: CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so
CGP_TRACE_E2E=1 prints stage / queue / device / D2H / copy-out laps of cgp_fit_predict_batch to stderr)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from corenav_gp_amd import engine, synth
kid, X, y, Xs, th, dts = synth.config(2, batch=512)
ctx = engine.Context(max_n=2048, max_m=599, max_d=6, max_batch=512)
for i in range(3):
    t0=time.perf_counter(); r = ctx.fit_predict_batch(X, y, Xs, th, kid); print("call ms", (time.perf_counter()-t0)*1e3, file=sys.stderr)
