#!/usr/bin/env python3
"""Randomised sweep of the batched stop-time look-ahead (k_lookahead, cgp_predict_stop_batch) against the host C++
path of the same ABI (cgp_predict_stop, itself pinned to the oracle by tests/test_host_abi.py): random horizon
lengths (0, 1, the reference's 599 / 748), slip means up to the clamp, sigma scales, filter snapshots, thresholds,
arrival / now offsets (incl. "late"), H packing flag, ensemble sizes.
   python tests/fuzz/fuzz_lookahead.py [seconds=30] [seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from corenav_gp_amd import engine as e, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = e.Context(max_n=8, max_m=8, max_d=1)
t_end, cases, bad = time.time() + budget, 0, 0
while time.time() < t_end:
    T = int(rng.choice([1, 2, 63, 64, 65, 200]))
    M = int(rng.choice([0, 1, 2, 5, 50, 599, 748]))
    thr = float(rng.choice([0.5, 3.0, 10.0]))
    bug = bool(rng.integers(0, 2))
    base = rng.uniform(-0.2, 0.9)
    means = np.clip(base + 0.1 * rng.normal(size=(T, max(M, 1))), -0.95, 0.95)[:, :M]
    sigmas = np.abs(rng.normal(0.05, 0.05, size=(T, max(M, 1))))[:, :M] * float(rng.choice([0.1, 1.0, 5.0]))
    states = [synth.filter_state(int(rng.integers(0, 1 << 20))) for _ in range(T)]
    P, Q, STM, Hv, pos = (np.stack([s[j] for s in states]) for j in range(5))
    scale = 10.0 ** rng.uniform(-3, 1, size=T)
    P, Q = P * scale[:, None], Q * scale[:, None]
    arrival = rng.uniform(0, 100, size=T)
    now = arrival + rng.choice([0.0, 0.2, 5.0, 1e6], size=T)
    fired, cmd, iout, xy = ctx.predict_stop_batch(means.reshape(T, M), sigmas.reshape(T, M), P, Q, STM, Hv, pos, arrival, now,
                                                  threshold=thr, h_bug_compatible=bug)
    cases += 1
    for k in range(T):
        hf, hc, hi, hxy = e.predict_stop(means[k], sigmas[k], P[k], Q[k], STM[k], Hv[k], pos[k], arrival[k], now[k],
                                         threshold=thr, h_bug_compatible=bug)
        ok = bool(fired[k]) == hf and iout[k] == hi and abs(cmd[k] - hc) <= 1e-12 * max(abs(hc), 1.0) and \
            abs(xy[k] - hxy) <= 1e-8 * max(abs(hxy), 1.0)   # ENU differences of ECEF coordinates ~ 6e6 m: 1e-8 m absolute
        if not ok:
            # a threshold crossing decided by the last bits of xy_err may land one step apart: accept only that
            near = abs(hxy - thr) <= 1e-7 * thr or abs(xy[k] - thr) <= 1e-7 * thr
            if not near:
                print("FAIL T", T, "M", M, "thr", thr, "bug", bug, "traj", k, (bool(fired[k]), cmd[k], iout[k], xy[k]), (hf, hc, hi, hxy))
                bad += 1
print(f"cases {cases} failures {bad}")
sys.exit(1 if bad else 0)
