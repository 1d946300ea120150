#!/usr/bin/env python3
"""Phase cycles of the diagonal-tile kernel (ablation build, CGP_DBG=512): debug helper of round 1."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import corenav_gp_amd.engine as e, corenav_gp_amd.synth as synth
kid, X, y, Xs, th, _ = synth.config(2, batch=8)
ctx = e.Context(max_n=2048, max_m=599, max_d=6, max_batch=8)
rc = ctx.fit_predict_batch(X, y, Xs, th, kid)
d = ctx.debug_read()
print("shader cycles (s_memtime): phaseA", d[0], "B", d[1], "C", d[2], "abc total", d[3], "inverse", d[4], "potf2+store", d[5], "mainloop+gram(last k)", d[6])
