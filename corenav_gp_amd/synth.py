"""Seeded synthetic slip windows.  The Pathfinder bag is not obtainable
(README.md:40 of the reference points at an external DOI), so every benchmark / parity input is
generated here; the only real series in the reference, core_navigation/script/slipVal.csv, is
carried as a data fixture under tests/golden/.

Feature channels follow what CoreNav publishes next to slip (`slip_cn_`, CoreNav.cpp:346): tick,
rear-wheel speed, INS forward speed, and for d = 6 three IMU channels (yaw rate, a_x, pitch).
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 20260
KERNEL_SE_ISO, KERNEL_SE_ARD, KERNEL_RBF_BROWNIAN = 0, 1, 2
ARD_ELL = (0.7, 1.1, 1.5, 2.0, 0.9, 1.3)


def _slip_series(rng, t):
    y = 0.1 * np.sin(2.0 * np.pi * t / 40.0)
    imp = rng.random(t.shape[0]) < 0.04
    y = y + 0.05 * imp + rng.normal(0.0, 0.03, t.shape[0])
    return np.clip(y, -0.999, 0.999)


def _features(rng, ticks, d):
    n = ticks.shape[0]
    cols = [ticks.astype(np.float64)]
    if d >= 2:
        cols.append(rng.uniform(0.6, 1.0, n))
    if d >= 3:
        cols.append(rng.uniform(0.5, 1.0, n))
    while len(cols) < d:
        cols.append(rng.normal(0.0, 1.0, n))
    return np.stack(cols, 1)


def window(N, d, M=599, seed=SEED_BASE, tick0=11):
    """One synthetic window: returns X (N,d), y (N,), Xs (M,d); inputs standardised with the
    training statistics (column-wise), ticks contiguous, test ticks follow the window."""
    rng = np.random.default_rng(seed)
    ticks = np.arange(tick0, tick0 + N + M)
    F = _features(rng, ticks, d)
    y = _slip_series(rng, ticks[:N].astype(np.float64))
    mu, sd = F[:N].mean(0), F[:N].std(0)
    F = (F - mu) / sd
    return np.ascontiguousarray(F[:N]), y, np.ascontiguousarray(F[N:])


def theta_for(kernel_id, d, y, rng=None):
    vy = float(np.var(y))
    if kernel_id == KERNEL_SE_ISO:
        return np.array([0.02, 1.0, 1e-3])
    if kernel_id == KERNEL_SE_ARD:
        ell = np.array(ARD_ELL[:d]) if rng is None else rng.uniform(0.5, 2.0, d)
        return np.concatenate([[1.7 * vy], ell, [0.05 * vy]])
    if kernel_id == KERNEL_RBF_BROWNIAN:
        return np.array([0.5, 30.0, 0.01, 0.002])
    raise ValueError(kernel_id)


def config(cfg, batch=1, N=None, M=599, first=0):
    """BASELINE.json configs -> (kernel_id, X[b,N,d], y[b,N], Xs[b,M,d], theta[b,nt], dtype); windows
    first .. first + batch - 1 of the config's sequence (a rank's shard of a sweep)."""
    if cfg == 1:
        kid, d, n, dt = KERNEL_SE_ISO, 3, N or 256, "f64"
    elif cfg == 2:
        kid, d, n, dt = KERNEL_SE_ARD, 6, N or 2048, "f64"
    elif cfg == 3:
        kid, d, n, dt = KERNEL_SE_ARD, 6, N or 1024, "f32"
    else:
        raise ValueError(cfg)
    Xs_, ys_, Xt_, th_ = [], [], [], []
    for b in range(first, first + batch):
        seed = SEED_BASE + cfg + 1000 * b
        X, y, Xt = window(n, d, M, seed)
        rng = np.random.default_rng(seed + 7) if (cfg == 3 or b > 0) else None
        th = theta_for(kid, d, y, rng if kid == KERNEL_SE_ARD else None)
        Xs_.append(X); ys_.append(y); Xt_.append(Xt); th_.append(th)
    return kid, np.stack(Xs_), np.stack(ys_), np.stack(Xt_), np.stack(th_), dt


def reference_window(n=149, tick0=11, seed=SEED_BASE):
    """A window shaped like the reference's real operating point (gp_slip_node.py:27-29,
    CoreNav.cpp:270-288): n <= 149 contiguous odometry ticks, d = 1."""
    rng = np.random.default_rng(seed)
    t = np.arange(tick0, tick0 + n, dtype=np.float64)
    return t, _slip_series(rng, t)


def filter_state(seed, with_H=False):
    """A plausible 15-state filter snapshot for the SetStopping response (P, Q, STM row-major 225
    each, HvecData[60] packed with the reference's r*4+c indexing, LLH position): position error
    grows through the velocity states so the 3 m horizontal threshold is crossed after some tens of
    seconds (gp_predictor.cpp:58-130)."""
    rng = np.random.default_rng(seed + 99)
    A = rng.normal(0, 1e-3, (15, 15))
    A[6:8, :] *= 1e-7
    A[:, 6:8] *= 1e-2
    STM = np.eye(15) + A * 0.02
    STM[6:9, 3:6] += np.diag([1.6e-9, 2.0e-9, -0.02])
    P = np.diag(np.concatenate([np.full(3, 1e-6), np.full(3, 2e-3), [1e-15, 1e-15, 0.04], np.full(6, 1e-8)]))
    Q = np.diag(np.concatenate([np.full(3, 1e-9), np.full(3, 3e-5), [4e-19, 6e-19, 1e-6], np.full(6, 1e-12)]))
    H = np.zeros((4, 15))
    H[0, 3], H[1, 2], H[2, 4], H[3, 5] = 1.0, 1.0, 1.0, 1.0
    H += rng.normal(0, 1e-3, H.shape)
    hvec = np.zeros(60)
    for r in range(4):                      # CoreNav.cpp:669-673 packing
        for c in range(15):
            hvec[r * 4 + c] = H[r, c]
    pos = np.array([0.693457963620326, -1.39498384275845, 334.993517334743]) + np.array([1e-6, -2e-6, 1.5])
    if with_H:   # the filter's own 4 x 15 measurement matrix (the packed form above loses most of it: SURVEY 8a quirks)
        return P.reshape(225), Q.reshape(225), STM.reshape(225), hvec, pos, H
    return P.reshape(225), Q.reshape(225), STM.reshape(225), hvec, pos
