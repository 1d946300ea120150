import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import corenav_gp_amd.engine as engine, corenav_gp_amd.synth as synth
from oracle import gp_oracle as go
B = 64
kid, X, y, Xs, th, _ = synth.config(3, batch=B)
print("shapes", X.shape, Xs.shape)
ctx = engine.Context(max_n=1024, max_m=599, max_d=6, max_batch=B, dtype=engine.F32)
for call in range(3):
    rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, kid)
    worst = []
    for b in range(0, B, 8):
        f = go.fit(kid, th[b], X[b], y[b])
        omu, ovar = go.predict(f, Xs[b])
        worst.append((b, float(np.max(np.abs(mean[b] - omu)) / np.max(np.abs(omu))), float(np.max(np.abs(var[b] - ovar) / ovar)), float(abs(logml[b] - f.logml) / abs(f.logml))))
    print("call", call, "rc", rc, "info", int(np.count_nonzero(info)), ["%d: %.1e %.1e %.1e" % w for w in worst])
