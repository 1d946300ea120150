"""ORACLE (test infrastructure, NOT product code) -- numpy/scipy fp64 restatement of the
corenav-GP time-series GP slip predictor and of the GpPredictor stop-time look-ahead.

PARITY UNPINNED at the GPy boundary: the reference's arithmetic lives in GPy (un-vendored, unpinned,
not importable here: `core_navigation/script/gp_slip_node.py:3`), and the reference holds no test or
golden vector for this path.  What pins this file instead:
  * closed-form known answers (N=1, N=2, noise->inf limit, Brownian prior variance) -- tests/test_oracle.py
  * an independent implementation, scikit-learn GaussianProcessRegressor(optimizer=None), for the
    SE-iso / SE-ARD kernels at fixed theta -- fixtures in tests/golden/ made by tests/golden/gen_golden.py
  * algebraic properties (L L^T = Ky, Ky alpha = y, var >= noise, permutation invariance).

Every function cites the reference file:line whose behaviour it restates.  Paths are relative to
/root/reference (the upstream repository); GPy internals are cited as documented in SURVEY.md 3B.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

# kernel ids -- shared with include/corenav_gp.h
KERNEL_SE_ISO = 0        # theta = [sigma_f^2, ell, sigma_n^2]
KERNEL_SE_ARD = 1        # theta = [sigma_f^2, ell_1..ell_d, sigma_n^2]
KERNEL_RBF_BROWNIAN = 2  # theta = [sigma_r^2, ell, sigma_b^2, sigma_n^2], d == 1 (reference kernel)

GPY_DIAG_EPS = 1e-8      # GPy ExactGaussianInference adds (noise + 1e-8) to diag(K)
GPY_VAR_FLOOR = 1e-15    # GPy Posterior._raw_predict clips the latent variance at 1e-15
LOG_2PI = math.log(2.0 * math.pi)


def n_theta(kernel_id: int, d: int) -> int:
    return {KERNEL_SE_ISO: 3, KERNEL_SE_ARD: d + 2, KERNEL_RBF_BROWNIAN: 4}[kernel_id]


def noise_var(kernel_id: int, theta) -> float:
    return float(theta[-1])


# --------------------------------------------------------------------------------------------------
# a2: kernel functions.  gp_slip_node.py:31 builds `GPy.kern.RBF(1) * GPy.kern.Brownian(1)`.
# --------------------------------------------------------------------------------------------------
def _gpy_rbf(X, X2, variance, lengthscale):
    """GPy Stationary._unscaled_dist + RBF.K_of_r: r^2 = x^2 + x'^2 - 2xx' clipped at
    0, diagonal forced to 0 when X2 is None, K = variance * exp(-0.5 r^2 / ell^2)."""
    if X2 is None:
        Xsq = np.sum(np.square(X), 1)
        r2 = -2.0 * X @ X.T + (Xsq[:, None] + Xsq[None, :])
        r2[np.diag_indices_from(r2)] = 0.0
    else:
        X1sq = np.sum(np.square(X), 1)
        X2sq = np.sum(np.square(X2), 1)
        r2 = -2.0 * X @ X2.T + (X1sq[:, None] + X2sq[None, :])
    r2 = np.clip(r2, 0.0, np.inf)
    r = np.sqrt(r2) / lengthscale
    return variance * np.exp(-0.5 * r ** 2)


def _gpy_brownian(X, X2, variance):
    """GPy Brownian.K: variance * min(|x|,|x'|) where sign(x) == sign(x'), else 0 (1-D only)."""
    if X2 is None:
        X2 = X
    return np.where(np.sign(X) == np.sign(X2.T), variance * np.fmin(np.abs(X), np.abs(X2.T)), 0.0)


def _se(X, X2, sigma_f2, ell):
    """SE / ARD kernel as scikit-learn evaluates it (ConstantKernel * RBF): squared euclidean
    distance of the length-scaled inputs by direct differences, K = sigma_f^2 exp(-0.5 d^2)."""
    A = X / ell
    B = A if X2 is None else X2 / ell
    diff = A[:, None, :] - B[None, :, :]
    d2 = np.einsum("ijk,ijk->ij", diff, diff)
    return sigma_f2 * np.exp(-0.5 * d2)


def kernel_K(kernel_id, theta, X, X2=None):
    """Cross-/auto-covariance K(X, X2) WITHOUT the noise term.  X is (n, d)."""
    X = np.asarray(X, dtype=np.float64)
    X2 = None if X2 is None else np.asarray(X2, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64)
    d = X.shape[1]
    if kernel_id == KERNEL_SE_ISO:
        return _se(X, X2, theta[0], theta[1])
    if kernel_id == KERNEL_SE_ARD:
        return _se(X, X2, theta[0], theta[1:1 + d])
    if kernel_id == KERNEL_RBF_BROWNIAN:
        if d != 1:
            raise ValueError("RBF x Brownian is defined for d == 1 (gp_slip_node.py:19-21,31)")
        return _gpy_rbf(X, X2, theta[0], theta[1]) * _gpy_brownian(X, X2, theta[2])
    raise ValueError(f"unknown kernel id {kernel_id}")


def kernel_Kdiag(kernel_id, theta, X):
    """Prior variance k(x*, x*) (GPy Kdiag): SE -> sigma_f^2; RBF x Brownian -> s_r^2 s_b^2 |x|."""
    X = np.asarray(X, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64)
    if kernel_id in (KERNEL_SE_ISO, KERNEL_SE_ARD):
        return np.full(X.shape[0], theta[0])
    if kernel_id == KERNEL_RBF_BROWNIAN:
        return theta[0] * theta[2] * np.abs(X[:, 0])
    raise ValueError(f"unknown kernel id {kernel_id}")


# --------------------------------------------------------------------------------------------------
# a3: jitchol.  GPy util.linalg.jitchol: dpotrf lower; on failure retry <= 5 times
# with jitter mean(diag) * 1e-6 * 10^k added to the diagonal.
# --------------------------------------------------------------------------------------------------
class NotPositiveDefinite(np.linalg.LinAlgError):
    pass


def jitchol(A, maxtries=5):
    A = np.ascontiguousarray(A)
    L, info = sla.lapack.dpotrf(A, lower=1)
    if info == 0:
        return np.tril(L), 0.0, 0
    diagA = np.diag(A)
    if np.any(diagA <= 0.0):
        raise NotPositiveDefinite("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        L, info = sla.lapack.dpotrf(A + np.eye(A.shape[0]) * jitter, lower=1)
        if info == 0:
            return np.tril(L), jitter, num_tries
        jitter *= 10.0
        num_tries += 1
    raise NotPositiveDefinite("not positive definite, even with jitter.")


# --------------------------------------------------------------------------------------------------
# a3-a6: exact Gaussian inference at FIXED theta (GPy ExactGaussianInference.inference; the
# reference reaches it through GPRegression(...) gp_slip_node.py:35 and m.optimize() :36).
# --------------------------------------------------------------------------------------------------
class Fit:
    __slots__ = ("kernel_id", "theta", "X", "y", "L", "alpha", "z", "logml", "jitter", "Kyinv")


def fit(kernel_id, theta, X, y, want_inverse=False) -> Fit:
    """Ky = K + (sigma_n^2 + 1e-8) I ; L = jitchol(Ky) ; alpha = Ky^-1 y (dpotrs) ;
    logML = 0.5 (-N log 2pi - 2 sum log L_ii - y^T alpha)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    N = X.shape[0]
    K = kernel_K(kernel_id, theta, X)
    Ky = K.copy()
    Ky[np.diag_indices(N)] += noise_var(kernel_id, theta) + GPY_DIAG_EPS
    L, jitter, _ = jitchol(Ky)
    z = sla.solve_triangular(L, y, lower=True)            # L z = y
    alpha = sla.solve_triangular(L, z, lower=True, trans="T")  # L^T alpha = z  (dpotrs)
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    f = Fit()
    f.kernel_id, f.theta, f.X, f.y = kernel_id, np.asarray(theta, dtype=np.float64), X, y
    f.L, f.alpha, f.z, f.jitter = L, alpha, z, jitter
    f.logml = 0.5 * (-N * LOG_2PI - logdet - float(y @ alpha))
    f.Kyinv = None
    if want_inverse:  # GPy pdinv -> dpotri (a4); only used by the variance cross-check
        Ai, _ = sla.lapack.dpotri(L, lower=1)
        f.Kyinv = np.tril(Ai) + np.tril(Ai, -1).T
    return f


# --------------------------------------------------------------------------------------------------
# a8: prediction.  gp_slip_node.py:45-49 calls m.predict([[x]]) one point at a time; GP.predict
# -> Posterior._raw_predict: mu = k*^T alpha ; var = k** - k*^T Ky^-1 k*, clipped >= 1e-15 ;
# include_likelihood=True adds sigma_n^2.
# --------------------------------------------------------------------------------------------------
def predict(f: Fit, Xs, include_noise=True, via_inverse=False):
    Xs = np.asarray(Xs, dtype=np.float64)
    if Xs.ndim == 1:
        Xs = Xs[:, None]
    Ks = kernel_K(f.kernel_id, f.theta, f.X, Xs)         # (N, M)
    mu = Ks.T @ f.alpha
    kss = kernel_Kdiag(f.kernel_id, f.theta, Xs)
    if via_inverse:                                        # literal GPy form (uses dpotri inverse)
        var = kss - np.sum((f.Kyinv.T @ Ks) * Ks, 0)
    else:                                                  # algebraically identical, trsm form
        V = sla.solve_triangular(f.L, Ks, lower=True)
        var = kss - np.sum(V * V, 0)
    var = np.clip(var, GPY_VAR_FLOOR, np.inf)
    if include_noise:
        var = var + noise_var(f.kernel_id, f.theta)
    return mu, var


# --------------------------------------------------------------------------------------------------
# a1 + a9: the node callback around the fit (gp_slip_node.py:16-63) at FIXED theta.
# --------------------------------------------------------------------------------------------------
TRAIN_FRACTION = 0.9     # gp_slip_node.py:27
HORIZON_TICKS = 600      # gp_slip_node.py:45


def slip_node_split(time_array, slip_array):
    """gp_slip_node.py:19-30: (n,1) reshape, first int(0.9 n) samples train, rest unused."""
    X = np.asarray(time_array, dtype=np.float64).reshape(-1, 1)
    Y = np.asarray(slip_array, dtype=np.float64).reshape(-1, 1)
    ntr = int(TRAIN_FRACTION * len(X))
    return X, Y, X[:ntr], Y[:ntr]


def slip_node_grid(X):
    """gp_slip_node.py:45: X_ = arange(X.min(), X.max() + 600, 1)."""
    return np.arange(X.min(), X.max() + HORIZON_TICKS, 1)


def slip_node_callback(time_array, slip_array, theta, kernel_id=KERNEL_RBF_BROWNIAN):
    """Returns (mean, sigma) exactly as published on gp_result (gp_slip_node.py:57-63):
    mean = means[len(X):], sigma = 2 sqrt(var[len(X):]) -- INDEX-sliced, not time-sliced."""
    X, Y, xtr, ytr = slip_node_split(time_array, slip_array)
    f = fit(kernel_id, theta, xtr, ytr[:, 0])
    X_ = slip_node_grid(X)
    mu, var = predict(f, X_[:, None], include_noise=True)
    n = len(X)
    return mu[n:], 2.0 * np.sqrt(var[n:])


# --------------------------------------------------------------------------------------------------
# a11: GpPredictor::llh_to_enu (gp_predictor/src/gp_predictor.cpp:144-178).
# --------------------------------------------------------------------------------------------------
WGS84_A = 6378137.0
WGS84_B = 6356752.3142
# core_navigation/config/init_params.yaml:9-16 (the reference never loads them: LoadParameters is
# not called, gp_predictor.cpp:134-142; the restatement loads them so the rotation is defined).
INIT_LLH = (0.693457963620326, -1.39498384275845, 334.993517334743)
INIT_ECEF = (859153.0153, -4836303.7266, 4055378.501)


def llh_to_enu(lat, lon, h, init_llh=INIT_LLH, init_ecef=INIT_ECEF):
    a, b = WGS84_A, WGS84_B
    e = math.sqrt(1.0 - (b / a) ** 2)
    sinphi, cosphi = math.sin(lat), math.cos(lat)
    coslam, sinlam = math.cos(lon), math.sin(lon)
    tan2phi = math.tan(lat) ** 2
    tmp2 = 1.0 - e * e
    tmpden = math.sqrt(1.0 + tmp2 * tan2phi)
    x1 = (a * coslam) / tmpden + h * coslam * cosphi
    y1 = (a * sinlam) / tmpden + h * sinlam * cosphi
    tmp3 = math.sqrt(1.0 - e * e * sinphi * sinphi)
    z1 = (a * tmp2 * sinphi) / tmp3 + h * sinphi
    d = np.array([x1 - init_ecef[0], y1 - init_ecef[1], z1 - init_ecef[2]])
    sP, cP = math.sin(init_llh[0]), math.cos(init_llh[0])
    sL, cL = math.sin(init_llh[1]), math.cos(init_llh[1])
    R = np.array([[-sL, cL, 0.0], [-sP * cL, -sP * sL, cP], [cP * cL, cP * sL, sP]])
    return R @ d


# --------------------------------------------------------------------------------------------------
# a12: SetStopping (un)packing incl. the reference's H indexing quirk
# (CoreNav.cpp:652-676 server side, gp_predictor.cpp:30-46 client side).
# --------------------------------------------------------------------------------------------------
def pack_set_stopping(P, Q, STM, H, pos):
    res = {"PvecData": np.asarray(P).reshape(225).copy(), "QvecData": np.asarray(Q).reshape(225).copy(),
           "STMvecData": np.asarray(STM).reshape(225).copy(), "HvecData": np.zeros(60),
           "PosData": np.asarray(pos, dtype=np.float64).copy()}
    H = np.asarray(H)
    for r in range(4):            # CoreNav.cpp:669-673 writes index r*4+c for a 4x15 matrix
        for c in range(15):
            res["HvecData"][r * 4 + c] = H[r, c]
    return res


def unpack_H(HvecData, bug_compatible=True):
    H = np.zeros((4, 15))
    for r in range(4):            # gp_predictor.cpp:38-42 reads index r*4+c
        for c in range(15):
            H[r, c] = HvecData[r * 4 + c] if bug_compatible else HvecData[r * 15 + c]
    return H


# --------------------------------------------------------------------------------------------------
# a10: GpPredictor::GPCallBack covariance look-ahead (gp_predictor.cpp:58-130).
# --------------------------------------------------------------------------------------------------
def predict_stop(mean, sigma, P, Q, STM, H, pos_llh, arrival_time=0.0, now=0.0,
                 threshold=3.0, init_llh=INIT_LLH, init_ecef=INIT_ECEF, return_trace=False):
    """Returns (fired, stop_cmd, i, xy_err[, trace]).  stop_cmd is the Float64 published on stop_cmd
    (gp_predictor.cpp:107-118); fired is False when the loop ends without crossing the threshold."""
    mean = np.asarray(mean, dtype=np.float64)
    sigma = np.asarray(sigma, dtype=np.float64)
    P = np.array(P, dtype=np.float64).reshape(15, 15)
    Q = np.asarray(Q, dtype=np.float64).reshape(15, 15)
    STM = np.asarray(STM, dtype=np.float64).reshape(15, 15)
    H = np.asarray(H, dtype=np.float64).reshape(4, 15)
    I15 = np.eye(15)
    R1 = np.array([[0.5, 0.5, 0.0, 0.0], [1 / 0.685, -1 / 0.685, 0.0, 0.0],
                   [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])        # :84-87
    i = 0
    xy = 0.0
    trace = []
    for slip_i in range(5 * len(mean)):                                 # :64
        P = STM @ P @ STM.T + Q                                         # :66
        if slip_i % 5 == 0:                                             # :67
            c0 = mean[i]
            c1 = mean[i] + sigma[i]
            c2 = mean[i] - sigma[i]
            o0, o1, o2 = 0.8 / (1.0 - c0), 0.8 / (1.0 - c1), 0.8 / (1.0 - c2)   # :73-75
            est = (o0 + o1 + o2) / 3.0
            cov = ((o0 - est) ** 2 + (o1 - est) ** 2 + (o2 - est) ** 2) / 3.0    # :78
            R2 = np.diag([max(0.03 * 0.03, cov * cov), max(0.03 * 0.03, cov * cov),
                          max(0.05 * 0.05, cov * cov), 0.05 * 0.05])              # :80-83
            R = 25.0 * R1 @ R2 @ R1.T                                             # :88
            K = P @ H.T @ np.linalg.inv(H @ P @ H.T + R)                          # :90
            IKH = I15 - K @ H
            P = IKH @ P @ IKH.T + K @ R @ K.T                                     # :91
            i += 1                                                                # :92
        s6, s7, s8 = (3.0 * math.sqrt(abs(P[6, 6])), 3.0 * math.sqrt(abs(P[7, 7])),
                      3.0 * math.sqrt(abs(P[8, 8])))
        e0 = llh_to_enu(pos_llh[0], pos_llh[1], pos_llh[2], init_llh, init_ecef)        # :95
        e3 = llh_to_enu(pos_llh[0] + s6, pos_llh[1] + s7, pos_llh[2] + s8, init_llh, init_ecef)  # :97
        xy = math.sqrt((e3[0] - e0[0]) ** 2 + (e3[1] - e0[1]) ** 2)              # :99
        if return_trace:
            trace.append(xy)
        if xy > threshold:                                                        # :102
            dt = arrival_time + i / 10.0 - now                                    # :107,114
            cmd = 0.5 if dt < 0.0 else dt
            return (True, cmd, i, xy, np.array(trace)) if return_trace else (True, cmd, i, xy)
    return (False, 0.0, i, xy, np.array(trace)) if return_trace else (False, 0.0, i, xy)


# --------------------------------------------------------------------------------------------------
# cfg4: sliding-window rank-1 Cholesky up/down-date (BASELINE.json configs[3]; NOT in the reference,
# which refits once per window: CoreNav.cpp:289-305).  Restated as plain refits: the oracle for a
# streamed window is "fit from scratch on the current window".
# --------------------------------------------------------------------------------------------------
def sliding_window_refit(kernel_id, theta, Xwin, ywin):
    return fit(kernel_id, theta, Xwin, ywin)


def sliding_window_stream(kernel_id, theta, N, xs, ys, include_noise=True):
    """Tick-by-tick oracle of the sliding window: before a sample enters, predict it from the current
    window (one-step-ahead mean / variance); then drop the oldest sample if the window is full, add
    the new one and REFIT from scratch.  Returns (pred_mean, pred_var, logml) per tick."""
    xs = np.asarray(xs, dtype=np.float64)
    if xs.ndim == 1:
        xs = xs[:, None]
    ys = np.asarray(ys, dtype=np.float64)
    T = len(ys)
    pm, pv, lm = np.zeros(T), np.zeros(T), np.zeros(T)
    Xw, yw = np.zeros((0, xs.shape[1])), np.zeros(0)
    for t in range(T):
        if len(yw) >= N:                      # the oldest sample leaves BEFORE the prediction of the new one
            Xw, yw = Xw[1:], yw[1:]
        if len(yw) == 0:
            pm[t] = 0.0
            v = kernel_Kdiag(kernel_id, theta, xs[t:t + 1])[0]
            pv[t] = v + (noise_var(kernel_id, theta) if include_noise else 0.0)
        else:
            f = fit(kernel_id, theta, Xw, yw)
            mu, var = predict(f, xs[t:t + 1], include_noise)
            pm[t], pv[t] = mu[0], var[0]
        Xw, yw = np.vstack([Xw, xs[t:t + 1]]), np.append(yw, ys[t])
        lm[t] = fit(kernel_id, theta, Xw, yw).logml
    return pm, pv, lm


# --------------------------------------------------------------------------------------------------
# a7 / f2: hyper-parameter optimisation.  gp_slip_node.py:36 `m.optimize()` = paramz 'lbfgsb'
# (scipy fmin_l_bfgs_b, no bounds, maxfun 1000) on the Logexp-transformed parameters, objective
# -log p(y|X,theta), gradient from dL/dK = 0.5 (alpha alpha^T - Ky^-1)  (SURVEY.md 3B).
# --------------------------------------------------------------------------------------------------
def dK_dtheta(kernel_id, theta, X):
    """List of dK/dtheta_p (N x N each), natural parameters, noise last."""
    X = np.asarray(X, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64)
    N, d = X.shape
    K = kernel_K(kernel_id, theta, X)
    out = []
    if kernel_id in (KERNEL_SE_ISO, KERNEL_SE_ARD):
        ell = np.full(d, theta[1]) if kernel_id == KERNEL_SE_ISO else theta[1:1 + d]
        out.append(K / theta[0])
        dq2 = [((X[:, None, q] - X[None, :, q]) / ell[q]) ** 2 for q in range(d)]
        if kernel_id == KERNEL_SE_ISO:
            out.append(K * sum(dq2) / theta[1])
        else:
            out.extend(K * dq2[q] / ell[q] for q in range(d))
    else:
        Xsq = np.sum(np.square(X), 1)
        r2 = -2.0 * X @ X.T + (Xsq[:, None] + Xsq[None, :])
        r2[np.diag_indices_from(r2)] = 0.0
        r2 = np.clip(r2, 0.0, np.inf) / theta[1] ** 2
        out.append(K / theta[0])
        out.append(K * r2 / theta[1])
        out.append(K / theta[2])
    out.append(np.eye(N))
    return out


def nll_and_grad(kernel_id, theta, X, y):
    """Negative log marginal likelihood and its gradient wrt the natural parameters."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    f = fit(kernel_id, theta, X, y, want_inverse=True)
    dL_dK = 0.5 * (np.outer(f.alpha, f.alpha) - f.Kyinv)
    g = np.array([np.sum(dL_dK * dK) for dK in dK_dtheta(kernel_id, theta, X)])
    return -f.logml, -g


def logexp(x):
    return np.where(x > 35.0, x, np.log1p(np.exp(np.minimum(x, 35.0))))


def logexp_inv(th):
    return np.where(th > 35.0, th, np.log(np.expm1(np.minimum(th, 35.0))))


def optimize(kernel_id, X, y, theta0=None, max_evals=1000):
    """Returns (theta_opt, logml, n_evals); theta0 defaults to GPy's all-ones start."""
    import scipy.optimize as so
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    nth = n_theta(kernel_id, X.shape[1])
    th0 = np.ones(nth) if theta0 is None else np.asarray(theta0, dtype=np.float64)
    count = [0]

    def fg(x):
        count[0] += 1
        th = logexp(x)
        try:
            nll, g = nll_and_grad(kernel_id, th, X, y)
        except np.linalg.LinAlgError:
            return 1e300, np.zeros_like(x)
        return nll, g * -np.expm1(-th)        # dtheta/dx = 1 - exp(-theta)

    x, fval, _ = so.fmin_l_bfgs_b(fg, logexp_inv(th0), maxfun=max_evals)
    return logexp(x), -fval, count[0]


# --------------------------------------------------------------------------------------------------
# f4: producer side.  CoreNav::Update slip + recording-window state machine
# (core_navigation/src/CoreNav.cpp:176,244-330), stopCallback (:755-759), getCmdData (:794-816).
# --------------------------------------------------------------------------------------------------
class SlipRecorderOracle:
    def __init__(self):
        self.odomUptCount = 0.0
        self.first_driving_flag, self.gp_flag = True, False
        self.new_stop_data_arrived_, self.started_driving_again_flag = False, True
        self.startRecording = self.stopRecording = self.saveCountOdom = 0.0
        self.cmd_stop_ = 0.0
        self.skipped_windows = 0
        self.t, self.s = [], []
        self.slip = 0.0

    def update(self, vfl, vfr, vbl, vbr, vlin, cmd_x):
        """Returns (time_array, slip_array) when a window is published, else None."""
        out = None
        self.odomUptCount += 1                                                       # :176
        rear = (vbl + vbr) / 2.0
        with np.errstate(divide="ignore", invalid="ignore"):
            cands = [np.float64(v - vlin) / np.float64(v) for v in (vfr, vbr, vfl, vbl)]
        slip = max(max(cands[0], cands[1]), max(cands[2], cands[3]))                 # :246 (std::max, NaN-order kept)
        if abs(rear) < 0.001:
            slip = 0.0
        slip = min(max(slip, -1.0), 1.0) if slip == slip else slip
        self.slip = float(slip)
        if slip != 0.0 and slip != -1.0 and slip != 1.0 and abs(cmd_x) > 0.2:        # :264
            if self.first_driving_flag:
                self.saveCountOdom = self.odomUptCount
                self.startRecording = self.saveCountOdom + 10
                self.stopRecording = self.startRecording + 150
                self.first_driving_flag = False
            if self.startRecording < self.odomUptCount < self.stopRecording and not self.gp_flag:
                self.s.append(float(slip))
                self.t.append(self.odomUptCount)
            if self.odomUptCount >= self.stopRecording:
                if not self.gp_flag:
                    self.gp_flag = True
                    if len(self.s) < 15:
                        self.skipped_windows += 1
                    else:
                        out = (np.array(self.t), np.array(self.s))
                    self.t, self.s = [], []
                if self.new_stop_data_arrived_:
                    self.new_stop_data_arrived_ = False
                    self.startRecording = self.stopRecording + math.ceil(self.cmd_stop_) * 10 + 10 + 50
                    self.stopRecording = self.startRecording + 150
                    self.gp_flag = False
            if (not self.first_driving_flag) and self.odomUptCount / 10 - self.stopRecording / 10 > 10:
                self.t, self.s = [], []
                self.first_driving_flag = True
                self.gp_flag = False
        return out

    def stop_callback(self, cmd_stop):
        self.cmd_stop_ = cmd_stop
        self.new_stop_data_arrived_ = True

    def cmd_callback(self, cmd_x):
        if self.gp_flag and abs(cmd_x) < 0.0001:
            self.started_driving_again_flag = False
        if not self.started_driving_again_flag and abs(cmd_x) > 0.0001:
            self.started_driving_again_flag = True
            self.gp_flag = False
