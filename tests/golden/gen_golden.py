#!/usr/bin/env python3
"""Generates tests/golden/*.npz (run in the BUILD container only; needs scikit-learn and
/root/reference for slipVal.csv).  The fixtures are data: inputs + expected outputs.

Sources of the expected values, per fixture `source` field:
  sklearn  -- scikit-learn GaussianProcessRegressor(optimizer=None): an implementation independent
              of oracle/ (the reference cites scikit-learn kernels, `Kernel Selection/README.md:9`)
  closed   -- closed-form N=1 / N=2 answers evaluated with python floats
  kalman   -- the reference's kernel in its pure-Brownian limit (RBF length-scale 1e12: the RBF factor is 1 to
              1e-18) at the reference's operating size (N = 134, M = 599): Brownian motion observed in white noise
              is a scalar state-space model, so posterior mean / variance / log marginal likelihood come from a
              Kalman filter + RTS smoother recursion in python floats -- O(N), no Gram matrix, no Cholesky, nothing
              shared with oracle/ -- itself checked here against the textbook noise-free forms (linear
              interpolation between samples, constant beyond the last one, bridge variance s2 (x-a)(b-x)/(b-a),
              s2 (x - x_N) beyond)
  mpmath   -- the reference's product kernel RBF(1) x Brownian(1) (gp_slip_node.py:31) at a WORKING length-scale, on the
              training part of the slipVal window (N = 134) and the 599 published ticks (gp_slip_node.py:45-49,59-61):
              50-digit arithmetic (mpmath), kernel from its definition, LU with pivoting for the inverse and the
              determinant -- no Cholesky, no numpy, nothing of oracle/ -- posterior mean, variance, log marginal
              likelihood and its gradient wrt the four parameters
              mp_lookahead.npz: the GpPredictor look-ahead loop and llh_to_enu (gp_predictor.cpp:64-99,144-178) in the same
              50-digit arithmetic, from the C++ source, on the inputs of lookahead_restated.npz
  restated+scipy -- the reference's `m.optimize()` (gp_slip_node.py:36): scipy.optimize.fmin_l_bfgs_b (the optimiser GPy
              itself calls) on the restated objective from GPy's all-ones start; optimum, logML, evaluation count and the
              published (mean[599], sigma[599]) at the optimum
  restated -- oracle/gp_oracle.py outputs (regression vectors for the GPy-only RBF x Brownian kernel
              and for the GpPredictor look-ahead; PARITY UNPINNED vs GPy itself)
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as go  # noqa: E402
import corenav_gp_amd.synth as synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF_CSV = "/root/reference/core_navigation/script/slipVal.csv"


def sklearn_case(name, kid, X, y, Xs, theta):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    d = X.shape[1]
    ell = theta[1] if kid == go.KERNEL_SE_ISO else np.asarray(theta[1:1 + d])
    kern = ConstantKernel(theta[0], "fixed") * RBF(ell, "fixed")
    gpr = GaussianProcessRegressor(kern, alpha=theta[-1] + go.GPY_DIAG_EPS, optimizer=None,
                                   normalize_y=False).fit(X, y)
    mu, sd = gpr.predict(Xs, return_std=True)
    lml = gpr.log_marginal_likelihood_value_
    # cross-check the restatement against sklearn before writing anything
    f = go.fit(kid, theta, X, y)
    omu, ovar = go.predict(f, Xs, include_noise=False)
    e_mu = np.max(np.abs(omu - mu)) / max(np.max(np.abs(mu)), 1e-300)
    e_var = np.max(np.abs(ovar - sd ** 2) / np.maximum(sd ** 2, 1e-12))
    e_l = abs(f.logml - lml) / abs(lml)
    print(f"{name}: oracle vs sklearn  mean {e_mu:.2e}  var {e_var:.2e}  logml {e_l:.2e}")
    assert e_mu < 1e-9 and e_var < 1e-7 and e_l < 1e-11, name
    np.savez_compressed(os.path.join(OUT, name + ".npz"), source="sklearn", kernel_id=kid,
                        theta=np.asarray(theta, dtype=np.float64), X=X, y=y, Xs=Xs, mean=mu,
                        var_latent=sd ** 2, logml=lml, alpha=gpr.alpha_)


def closed_forms():
    # N = 1, SE-iso: mu = k* y/(s+n), var = s - k*^2/(s+n), logML = -0.5 y^2/(s+n) - 0.5 log(s+n) - 0.5 log 2pi
    s, ell, n = 1.7, 0.8, 0.3
    x, y, xs = 0.4, -0.6, 1.1
    neff = n + go.GPY_DIAG_EPS
    ks = s * math.exp(-0.5 * ((x - xs) / ell) ** 2)
    mu = ks * y / (s + neff)
    var = s - ks * ks / (s + neff)
    lml = -0.5 * y * y / (s + neff) - 0.5 * math.log(s + neff) - 0.5 * math.log(2 * math.pi)
    np.savez_compressed(os.path.join(OUT, "closed_n1_se.npz"), source="closed", kernel_id=0,
                        theta=np.array([s, ell, n]), X=np.array([[x]]), y=np.array([y]),
                        Xs=np.array([[xs]]), mean=np.array([mu]), var_latent=np.array([var]), logml=lml)
    # N = 2, RBF x Brownian (the reference kernel), explicit 2x2 inverse
    sr, l, sb, n = 0.9, 5.0, 0.05, 0.01
    x1, x2, xs = 12.0, 15.0, 20.0
    y1, y2 = 0.07, -0.02
    neff = n + go.GPY_DIAG_EPS
    k = lambda a, b: sr * math.exp(-0.5 * (a - b) ** 2 / l ** 2) * sb * min(abs(a), abs(b))
    a11, a12, a22 = k(x1, x1) + neff, k(x1, x2), k(x2, x2) + neff
    det = a11 * a22 - a12 * a12
    i11, i12, i22 = a22 / det, -a12 / det, a11 / det
    k1, k2 = k(x1, xs), k(x2, xs)
    mu = k1 * (i11 * y1 + i12 * y2) + k2 * (i12 * y1 + i22 * y2)
    var = k(xs, xs) - (k1 * (i11 * k1 + i12 * k2) + k2 * (i12 * k1 + i22 * k2))
    quad = y1 * (i11 * y1 + i12 * y2) + y2 * (i12 * y1 + i22 * y2)
    lml = -0.5 * quad - 0.5 * math.log(det) - math.log(2 * math.pi)
    np.savez_compressed(os.path.join(OUT, "closed_n2_rbfbrownian.npz"), source="closed", kernel_id=2,
                        theta=np.array([sr, l, sb, n]), X=np.array([[x1], [x2]]), y=np.array([y1, y2]),
                        Xs=np.array([[xs]]), mean=np.array([mu]), var_latent=np.array([var]), logml=lml)


def brownian_kalman(x, y, xs, s2, tau2):
    """Posterior of B(x*) for Brownian motion B (B(0) = 0, Var[B(b) - B(a)] = s2 (b - a)) observed as
    y_i = B(x_i) + N(0, tau2) at increasing x_i > 0.  Python floats only.  Returns mean[], latent var[], logML."""
    n = len(x)
    m, P, xp = 0.0, 0.0, 0.0
    fm, fP, pP = [], [], []          # filtered mean / variance, predicted variance at x_i
    lml = 0.0
    for i in range(n):
        Pm = P + s2 * (x[i] - xp)
        S = Pm + tau2
        K = Pm / S
        r = y[i] - m
        lml += -0.5 * (math.log(2.0 * math.pi * S) + r * r / S)
        m = m + K * r
        P = Pm * (1.0 - K)
        xp = x[i]
        fm.append(m); fP.append(P); pP.append(Pm)
    sm, sP = fm[:], fP[:]            # RTS smoother
    for i in range(n - 2, -1, -1):
        G = fP[i] / pP[i + 1]
        sm[i] = fm[i] + G * (sm[i + 1] - fm[i])
        sP[i] = fP[i] + G * G * (sP[i + 1] - pP[i + 1])
    mean, var = [], []
    for q in xs:
        if q >= x[-1]:
            mean.append(fm[-1]); var.append(fP[-1] + s2 * (q - x[-1]))
            continue
        j = 0
        while x[j] <= q:             # x[j] = first sample beyond q
            j += 1
        if j == 0:
            m0, P0, x0 = 0.0, 0.0, 0.0
        else:
            m0, P0, x0 = fm[j - 1], fP[j - 1], x[j - 1]
        Ps = P0 + s2 * (q - x0)      # filtered = predicted at the unobserved point q
        G = Ps / pP[j]
        mean.append(m0 + G * (sm[j] - m0))
        var.append(Ps + G * G * (sP[j] - pP[j]))
    return mean, var, lml


def brownian_cases():
    """The reference kernel `GPy.kern.RBF(1) * GPy.kern.Brownian(1)` (gp_slip_node.py:31) pinned without the oracle at
    the reference's operating size: first 134 ticks of the slipVal window as training set (gp_slip_node.py:27-29),
    (a) the 599 published prediction ticks (:45,:59 -- all beyond the window), (b) 599 points inside / before the
    window (bridge), (c) the prior-variance law: one sample, variance growing ~ x*."""
    raw = np.loadtxt(REF_CSV, delimiter=",")
    ticks = [float(round(t * 10.0)) for t in raw[:149, 0]]
    slip = [float(v) for v in raw[:149, 1]]
    ntr = int(0.9 * len(ticks))
    x, y = ticks[:ntr], slip[:ntr]
    sr, ell, sb, sn = 0.5, 1.0e12, 0.02, 0.002
    s2, tau2 = sr * sb, sn + go.GPY_DIAG_EPS
    # textbook check of the recursion itself (noise 1e-12 of the signal variance)
    xin = [x[0] * 0.5] + [x[i] + f for i in range(0, ntr - 1, 7) for f in (0.25, 0.5)] + [x[-1] + 40.0]
    km, kv, _ = brownian_kalman(x, y, xin, s2, 1e-14)
    for q, mq, vq in zip(xin, km, kv):
        if q < x[0]:
            em, ev = y[0] * q / x[0], s2 * q * (x[0] - q) / x[0]
        elif q > x[-1]:
            em, ev = y[-1], s2 * (q - x[-1])
        else:
            j = max(i for i in range(ntr) if x[i] <= q)
            a, b = x[j], x[j + 1]
            em = y[j] + (y[j + 1] - y[j]) * (q - a) / (b - a)
            ev = s2 * (q - a) * (b - q) / (b - a)
        assert abs(mq - em) < 1e-9 and abs(vq - ev) < 1e-9 * max(ev, 1e-3), (q, mq, em, vq, ev)
    theta = np.array([sr, ell, sb, sn])
    X = np.array(x)[:, None]
    cases = {
        "closed_brownian_kalman_n134": [ticks[0] + len(ticks) + m for m in range(599)],          # gp_slip_node.py:45,59
        "closed_brownian_bridge_n134": [x[0] * (0.1 + 0.8 * (m % 10) / 10.0) if m < 10 else
                                       x[(m * 7) % (ntr - 1)] + ((m * 37) % 100 + 1) / 101.0 for m in range(599)],
    }
    for name, xs in cases.items():
        mean, var, lml = brownian_kalman(x, y, xs, s2, tau2)
        f = go.fit(go.KERNEL_RBF_BROWNIAN, theta, X, np.array(y))
        omu, ovar = go.predict(f, np.array(xs)[:, None], include_noise=False)
        e_mu = np.max(np.abs(omu - mean)) / np.max(np.abs(mean))
        e_var = np.max(np.abs(ovar - var) / np.array(var))
        print(f"{name}: oracle vs Kalman/RTS  mean {e_mu:.2e}  var {e_var:.2e}  logml {abs(f.logml - lml) / abs(lml):.2e}")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), source="kalman", kernel_id=2, theta=theta, X=X,
                            y=np.array(y), Xs=np.array(xs)[:, None], mean=np.array(mean), var_latent=np.array(var), logml=lml)
    # (c) one sample at x = 1: var(x*) = P_1 + s2 (x* - 1), prior-like growth ~ x* up to tick 6000
    xs = [1.0 + 10.0 * m for m in range(1, 600)]
    mean, var, lml = brownian_kalman([1.0], [0.05], xs, s2, tau2)
    np.savez_compressed(os.path.join(OUT, "closed_brownian_prior_n1.npz"), source="kalman", kernel_id=2, theta=theta,
                        X=np.array([[1.0]]), y=np.array([0.05]), Xs=np.array(xs)[:, None], mean=np.array(mean),
                        var_latent=np.array(var), logml=lml)


def mp_rbfbrownian():
    """mp_rbfbrownian_n134.npz: independent pin of the reference's kernel at theta = (0.5, 30, 0.01, 0.002)."""
    import mpmath as mp
    mp.mp.dps = 50
    raw = np.loadtxt(REF_CSV, delimiter=",")
    ticks = np.round(raw[:, 0] * 10.0)[:149]
    slip = raw[:149, 1]
    n = len(ticks)
    ntr = int(0.9 * n)                                   # gp_slip_node.py:27-29
    xs_grid = np.arange(ticks.min(), ticks.max() + 600, 1)[n:]   # :45, :59-61 (index slice)
    x = [mp.mpf(float(v)) for v in ticks[:ntr]]
    y = [mp.mpf(float(v)) for v in slip[:ntr]]
    sr, ell, sb, sn = (mp.mpf(v) for v in ("0.5", "30", "0.01", "0.002"))

    def k(a, b):                                         # GPy RBF.K * Brownian.K for positive inputs
        return sr * mp.e ** (-(a - b) ** 2 / (2 * ell ** 2)) * sb * min(a, b)
    N = ntr
    K = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = k(x[i], x[j])
    Ky = K.copy()
    for i in range(N):
        Ky[i, i] += sn + mp.mpf("1e-8")                  # GPy ExactGaussianInference: K + (variance + 1e-8) I
    Kinv = mp.inverse(Ky)                                # LU with partial pivoting
    yv = mp.matrix(y)
    alpha = Kinv * yv
    P, L, U = mp.lu(Ky)
    logdet = sum(mp.log(abs(U[i, i])) for i in range(N))
    logml = -(yv.T * alpha)[0] / 2 - logdet / 2 - mp.mpf(N) / 2 * mp.log(2 * mp.pi)
    mean, var = [], []
    for xsv in xs_grid:
        a = mp.mpf(float(xsv))
        ks = mp.matrix([k(a, x[i]) for i in range(N)])
        mean.append((ks.T * alpha)[0])
        var.append(k(a, a) - (ks.T * (Kinv * ks))[0])    # latent variance (no noise term)
    W = alpha * alpha.T - Kinv                           # 2 dL/dK
    g = [mp.mpf(0)] * 4
    for i in range(N):
        for j in range(N):
            wk = W[i, j] * K[i, j]
            g[0] += wk / sr
            g[1] += wk * (x[i] - x[j]) ** 2 / ell ** 3
            g[2] += wk / sb
        g[3] += W[i, i]
    grad = [v / 2 for v in g]                            # d logML / d theta
    # the restatement against it, before anything is written
    theta = np.array([0.5, 30.0, 0.01, 0.002])
    xtr, ytr = ticks[:ntr, None], slip[:ntr]
    f = go.fit(go.KERNEL_RBF_BROWNIAN, theta, xtr, ytr)
    omu, ovar = go.predict(f, xs_grid[:, None], include_noise=False)
    onll, og = go.nll_and_grad(go.KERNEL_RBF_BROWNIAN, theta, xtr, ytr)
    mu = np.array([float(v) for v in mean])
    vr = np.array([float(v) for v in var])
    gr = np.array([float(v) for v in grad])
    e_mu = np.max(np.abs(omu - mu)) / np.max(np.abs(mu))
    e_var = np.max(np.abs(ovar - vr) / vr)
    e_l = abs(f.logml - float(logml)) / abs(float(logml))
    e_g = np.max(np.abs(-og - gr)) / np.max(np.abs(gr))
    print(f"mp_rbfbrownian_n134: oracle vs 50-digit LU  mean {e_mu:.2e}  var {e_var:.2e}  logml {e_l:.2e}  grad {e_g:.2e}")
    assert e_mu < 1e-9 and e_var < 1e-7 and e_l < 1e-11 and e_g < 1e-8
    np.savez_compressed(os.path.join(OUT, "mp_rbfbrownian_n134.npz"), source="mpmath", kernel_id=2, theta=theta, X=xtr, y=ytr,
                        Xs=xs_grid[:, None], mean=mu, var_latent=vr, logml=float(logml), dlogml_dtheta=gr,
                        alpha=np.array([float(v) for v in alpha]))


def slipval_window():
    """The only real slip series in the reference (core_navigation/script/slipVal.csv, 199 rows
    time_s, slip @ 0.1 s).  Stored as data: tick = round(10 t), the first 149 rows form one
    recording window (CoreNav.cpp:270-288)."""
    raw = np.loadtxt(REF_CSV, delimiter=",")
    ticks = np.round(raw[:, 0] * 10.0)
    slip = raw[:, 1]
    t, s = ticks[:149], slip[:149]
    theta = np.array([0.5, 30.0, 0.01, 0.002])
    mean, sigma = go.slip_node_callback(t, s, theta)
    X, Y, xtr, ytr = go.slip_node_split(t, s)
    f = go.fit(go.KERNEL_RBF_BROWNIAN, theta, xtr, ytr[:, 0])
    np.savez_compressed(os.path.join(OUT, "slipval_window_rbfbrownian.npz"), source="restated", kernel_id=2,
                        theta=theta, time_array=t, slip_array=s, mean=mean, sigma=sigma, logml=f.logml,
                        alpha=f.alpha, all_ticks=ticks, all_slip=slip)
    print("slipVal window: n", len(t), "ntrain", len(xtr), "M_out", len(mean), "logml", f.logml)


def restated_cases():
    t, s = synth.reference_window(149, tick0=11, seed=synth.SEED_BASE)
    theta = np.array([0.5, 30.0, 0.01, 0.002])
    mean, sigma = go.slip_node_callback(t, s, theta)
    np.savez_compressed(os.path.join(OUT, "synth_window_rbfbrownian.npz"), source="restated", kernel_id=2,
                        theta=theta, time_array=t, slip_array=s, mean=mean, sigma=sigma)
    # look-ahead golden (gp_predictor.cpp:58-130) on a synthetic but plausible filter state
    rng = np.random.default_rng(synth.SEED_BASE + 99)
    A = rng.normal(0, 1e-3, (15, 15))
    A[6:8, :] *= 1e-7                                      # lat/lon rows are in radians
    A[:, 6:8] *= 1e-2
    STM = np.eye(15) + A * 0.02
    STM[6:9, 3:6] += np.diag([1.6e-9, 2.0e-9, -0.02])     # position <- velocity coupling (rad, rad, m)
    P = np.diag(np.concatenate([np.full(3, 1e-6), np.full(3, 2e-3), [1e-15, 1e-15, 0.04], np.full(6, 1e-8)]))
    Q = np.diag(np.concatenate([np.full(3, 1e-9), np.full(3, 3e-5), [4e-19, 6e-19, 1e-6], np.full(6, 1e-12)]))
    H = np.zeros((4, 15))
    H[0, 3], H[1, 2], H[2, 4], H[3, 5] = 1.0, 1.0, 1.0, 1.0
    H += rng.normal(0, 1e-3, H.shape)
    pos = np.array(go.INIT_LLH) + np.array([1e-6, -2e-6, 1.5])
    packed = go.pack_set_stopping(P, Q, STM, H, pos)
    Hc = go.unpack_H(packed["HvecData"], True)
    fired, cmd, i, xy, trace = go.predict_stop(mean, sigma, P, Q, STM, Hc, pos, arrival_time=100.0, now=100.25,
                                               return_trace=True)
    print("look-ahead: fired", fired, "cmd", cmd, "i", i, "xy", xy)
    np.savez_compressed(os.path.join(OUT, "lookahead_restated.npz"), source="restated", mean=mean, sigma=sigma,
                        PvecData=packed["PvecData"], QvecData=packed["QvecData"], STMvecData=packed["STMvecData"],
                        HvecData=packed["HvecData"], PosData=packed["PosData"], H_true=H, H_client=Hc,
                        arrival_time=100.0, now=100.25, fired=fired, stop_cmd=cmd, i=i, xy_err=xy, trace=trace)
    e = go.llh_to_enu(*(np.array(go.INIT_LLH) + np.array([2e-6, 3e-6, 4.0])))
    np.savez_compressed(os.path.join(OUT, "llh_to_enu_restated.npz"), source="restated",
                        llh=np.array(go.INIT_LLH) + np.array([2e-6, 3e-6, 4.0]), enu=e)


def mp_lookahead():
    """mp_lookahead.npz: independent pin of the stop-time look-ahead (rows a10 / f3).  GpPredictor::GPCallBack's loop
    (gp_predictor/src/gp_predictor.cpp:64-99) and llh_to_enu (:144-178) restated from the C++ source in 50-digit
    arithmetic (mpmath: matrices of mpf, LU inverse of the 4 x 4 innovation covariance, mp trigonometry) on the inputs of
    lookahead_restated.npz -- nothing of oracle/ is used for the expected values.  Stored: the xy_err trace of every
    propagation step up to the 3.0 m threshold, and for a ladder of thresholds the step that crosses it, the odometry
    index i and the stop command (:102-118)."""
    import mpmath as mp
    mp.mp.dps = 50
    g = np.load(os.path.join(OUT, "lookahead_restated.npz"))
    F = lambda v: mp.mpf(float(v))
    M15 = lambda a: mp.matrix([[F(a[r * 15 + c]) for c in range(15)] for r in range(15)])
    P, Q, STM = M15(g["PvecData"]), M15(g["QvecData"]), M15(g["STMvecData"])
    H = mp.matrix(4, 15)
    for r in range(4):                       # gp_predictor.cpp:38-42 (row1*4+col1: the reference's indexing)
        for c in range(15):
            H[r, c] = F(g["HvecData"][r * 4 + c])
    pos = [F(v) for v in g["PosData"]]
    mean, sigma = [F(v) for v in g["mean"]], [F(v) for v in g["sigma"]]
    init_llh = [F(v) for v in go.INIT_LLH]    # core_navigation/config/init_params.yaml:9-16
    init_ecef = [F(v) for v in go.INIT_ECEF]

    def llh_to_enu(lat, lon, h):             # gp_predictor.cpp:144-178
        a, b = mp.mpf("6378137.0000"), mp.mpf("6356752.3142")
        e = mp.sqrt(1 - (b / a) ** 2)
        sinphi, cosphi, coslam, sinlam = mp.sin(lat), mp.cos(lat), mp.cos(lon), mp.sin(lon)
        tmp2 = 1 - e * e
        tmpden = mp.sqrt(1 + tmp2 * mp.tan(lat) ** 2)
        x1 = (a * coslam) / tmpden + h * coslam * cosphi
        y1 = (a * sinlam) / tmpden + h * sinlam * cosphi
        z1 = (a * tmp2 * sinphi) / mp.sqrt(1 - e * e * sinphi * sinphi) + h * sinphi
        dx, dy, dz = x1 - init_ecef[0], y1 - init_ecef[1], z1 - init_ecef[2]
        sP, cP, sL, cL = mp.sin(init_llh[0]), mp.cos(init_llh[0]), mp.sin(init_llh[1]), mp.cos(init_llh[1])
        return (-sL * dx + cL * dy, -sP * cL * dx - sP * sL * dy + cP * dz, cP * cL * dx + cP * sL * dy + sP * dz)

    R1 = mp.matrix([[0.5, 0.5, 0, 0], [1 / mp.mpf("0.685"), -1 / mp.mpf("0.685"), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    I15 = mp.eye(15)
    e0 = llh_to_enu(*pos)
    i, trace, istep = 0, [], []
    for slip_i in range(5 * len(mean)):      # :64
        P = STM * P * STM.T + Q              # :66
        if slip_i % 5 == 0:                  # :67
            o = [mp.mpf("0.8") / (1 - c) for c in (mean[i], mean[i] + sigma[i], mean[i] - sigma[i])]   # :69-75
            est = (o[0] + o[1] + o[2]) / 3
            cov = sum((v - est) ** 2 for v in o) / 3                                                  # :78
            f03, f05 = mp.mpf("0.03") ** 2, mp.mpf("0.05") ** 2
            R2 = mp.diag([max(f03, cov * cov), max(f03, cov * cov), max(f05, cov * cov), f05])         # :80-83
            R = 25 * R1 * R2 * R1.T                                                                    # :88
            K = P * H.T * mp.inverse(H * P * H.T + R)                                                  # :90
            IKH = I15 - K * H
            P = IKH * P * IKH.T + K * R * K.T                                                          # :91
            i += 1
        e3 = llh_to_enu(pos[0] + 3 * mp.sqrt(abs(P[6, 6])), pos[1] + 3 * mp.sqrt(abs(P[7, 7])), pos[2] + 3 * mp.sqrt(abs(P[8, 8])))
        xy = mp.sqrt((e3[0] - e0[0]) ** 2 + (e3[1] - e0[1]) ** 2)                                      # :99
        trace.append(xy)
        istep.append(i)
        if xy > 3:
            break
    trace_f = np.array([float(v) for v in trace])
    thresholds = np.array([0.9, 1.25, 1.5, 2.0, 2.5, 2.75, 3.0])
    cross = np.array([next(k for k, v in enumerate(trace) if v > mp.mpf(float(t))) for t in thresholds])
    arrival, now = float(g["arrival_time"]), float(g["now"])
    i_at = np.array([istep[k] for k in cross])
    stop_cmd = np.array([arrival + ii / 10.0 - now if arrival + ii / 10.0 - now >= 0 else 0.5 for ii in i_at])   # :107-118
    # the restatement against it, before anything is written
    fired, cmd, io, xyo, otrace = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"],
                                                  go.unpack_H(g["HvecData"], True), g["PosData"], arrival, now, return_trace=True)
    err = np.max(np.abs(otrace - trace_f) / trace_f)
    print(f"mp_lookahead: {len(trace_f)} steps, crossings {cross.tolist()}, i {i_at.tolist()}; oracle trace vs 50-digit: {err:.2e}")
    assert fired and io == i_at[-1] and len(otrace) == len(trace_f) and err < 1e-7
    np.savez_compressed(os.path.join(OUT, "mp_lookahead.npz"), source="mpmath", mean=g["mean"], sigma=g["sigma"],
                        PvecData=g["PvecData"], QvecData=g["QvecData"], STMvecData=g["STMvecData"], HvecData=g["HvecData"],
                        PosData=g["PosData"], arrival_time=arrival, now=now, trace=trace_f, thresholds=thresholds,
                        cross_step=cross, i_at=i_at, xy_at=trace_f[cross], stop_cmd=stop_cmd)


def optimised_theta_cases():
    """SURVEY 8c, last row: what the reference node publishes for a window WITH `m.optimize()` (gp_slip_node.py:31-36,
    45-61): all four parameters start at 1.0 (GPy defaults), paramz 'lbfgsb' = scipy.optimize.fmin_l_bfgs_b on the
    Logexp-transformed parameters (factr 1e7, pgtol 1e-5, maxfun 1000), then the 599 published ticks at the optimum.
    Expected values: oracle.optimize (scipy's own L-BFGS-B driving the restated objective) -- source "restated+scipy";
    the trajectory length (n_evals) is scipy's."""
    raw = np.loadtxt(REF_CSV, delimiter=",")
    windows = {"slipval_window_opt": (np.round(raw[:149, 0] * 10.0), raw[:149, 1]),
               "synth_window_opt": synth.reference_window(149, tick0=11, seed=synth.SEED_BASE)}
    for name, (t, s) in windows.items():
        X, Y, xtr, ytr = go.slip_node_split(t, s)
        th, lml, nev = go.optimize(go.KERNEL_RBF_BROWNIAN, xtr, ytr[:, 0])
        mean, sigma = go.slip_node_callback(t, s, th)
        f = go.fit(go.KERNEL_RBF_BROWNIAN, th, xtr, ytr[:, 0])
        assert abs(f.logml - lml) <= 1e-9 * abs(lml)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), source="restated+scipy", kernel_id=2, theta0=np.ones(4),
                            theta=th, logml=lml, n_evals=nev, time_array=t, slip_array=s, mean=mean, sigma=sigma)
        print(f"{name}: theta {th}  logml {lml:.12g}  evaluations {nev}  M_out {len(mean)}")
    # an SE-ARD window (d = 3) for the large-window machinery's optimiser (host L-BFGS over device gradients)
    Xw, yw, Xs = synth.window(192, 3, 16, seed=4242)
    th, lml, nev = go.optimize(go.KERNEL_SE_ARD, Xw, yw, np.ones(5))
    f = go.fit(go.KERNEL_SE_ARD, th, Xw, yw)
    mu, var = go.predict(f, Xs)
    np.savez_compressed(os.path.join(OUT, "synth_se_ard_n192_d3_opt.npz"), source="restated+scipy", kernel_id=1,
                        theta0=np.ones(5), theta=th, logml=lml, n_evals=nev, X=Xw, y=yw, Xs=Xs, mean=mu, var=var)
    print(f"synth_se_ard_n192_d3_opt: theta {th}  logml {lml:.12g}  evaluations {nev}")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-opt" in sys.argv:
        return optimised_theta_cases()
    if "--only-mp-lookahead" in sys.argv:
        return mp_lookahead()
    closed_forms()
    # cfg1-like (N=256,d=3 SE-iso) and smaller ARD cases, all against scikit-learn
    kid, X, y, Xs, th, _ = synth.config(1, M=64)
    sklearn_case("sk_se_iso_n256_d3", kid, X[0], y[0], Xs[0], th[0])
    for n, d, m, tag in [(2, 1, 5, "n2_d1"), (15, 3, 9, "n15_d3"), (134, 6, 48, "n134_d6"), (256, 6, 64, "n256_d6")]:
        Xn, yn, Xsn = synth.window(n, d, m, seed=synth.SEED_BASE + 10 * n + d)
        theta = synth.theta_for(go.KERNEL_SE_ARD, d, yn if n > 2 else np.array([0.3, -0.2]))
        sklearn_case(f"sk_se_ard_{tag}", go.KERNEL_SE_ARD, Xn, yn, Xsn, theta)
    # BASELINE configs[1] at full size (N=2048, d=6 ARD, M=599): the headline workload pinned by an
    # independent implementation
    kid, X, y, Xs, th, _ = synth.config(2)
    sklearn_case("sk_se_ard_n2048_d6", kid, X[0], y[0], Xs[0], th[0])
    brownian_cases()
    if "--no-mp" not in sys.argv:
        mp_rbfbrownian()
    slipval_window()
    restated_cases()
    if "--no-mp" not in sys.argv:
        mp_lookahead()
    optimised_theta_cases()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot)


if __name__ == "__main__":
    main()
