#!/usr/bin/env python3
"""fp32 mid-size call: extra rows as 64-row tiles on the second stream (k_rows64) against the one-stream 128-row form --
bitwise (measurement library: CGP_XSPLIT32=0 forces the old form).  python tools/xsplit32_bitwise.py [batch] [M]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 599
code = f"""
import sys, numpy as np; sys.path.insert(0, {ROOT!r})
import torch
from corenav_gp_amd import engine, synth
kid, X, y, Xs, th, _ = synth.config(3, batch={batch}, M={M})
ctx = engine.Context(max_n=X.shape[1], max_m={M}, max_d=6, max_batch={batch}, dtype=engine.F32)
for rep in range(2):
    rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, kid)
assert rc == 0 and not info.any()
np.save(sys.argv[1], np.concatenate([mean.ravel(), var.ravel(), logml]))
"""
ab = os.path.join(ROOT, "corenav_gp_amd", "libcorenav_gp_ab.so")
outs = []
for x in ("0", "1"):
    f = f"/tmp/xs32_{x}.npy"
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, CGP_LIB=ab, CGP_XSPLIT32=x), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    import numpy as np
    outs.append(np.load(f))
same = bool(np.array_equal(outs[0], outs[1]))
print(json.dumps({"batch": batch, "M": M, "bitwise_equal": same, "max_abs_diff": float(np.max(np.abs(outs[0] - outs[1])))}))
sys.exit(0 if same else 1)
