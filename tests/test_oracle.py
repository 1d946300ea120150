"""CPU tests: the oracle (numpy + C restatements) against the committed golden vectors, closed
forms and algebraic properties.  The reference holds no test on this path, so
these are the pins the parity claims rest on."""
import math

import numpy as np
import pytest

import os

from conftest import GOLDEN, load_golden
from oracle import gp_oracle as go

SK = ["sk_se_iso_n256_d3", "sk_se_ard_n2_d1", "sk_se_ard_n15_d3", "sk_se_ard_n134_d6", "sk_se_ard_n256_d6",
      "sk_se_ard_n2048_d6"]
CLOSED = ["closed_n1_se", "closed_n2_rbfbrownian"]
# RBF x Brownian (the reference's kernel, gp_slip_node.py:31) with the RBF factor switched off (ell = 1e12), at the
# reference's operating size: expected values from a Kalman filter / RTS smoother recursion in python floats
# (tests/golden/gen_golden.py: brownian_cases), nothing shared with oracle/
KALMAN = ["closed_brownian_kalman_n134", "closed_brownian_bridge_n134", "closed_brownian_prior_n1"]
# the reference's product kernel at a working length-scale (theta = 0.5, 30, 0.01, 0.002), N = 134, M = 599: 50-digit LU
# with pivoting (mpmath), kernel from its definition -- tests/golden/gen_golden.py: mp_rbfbrownian; nothing of oracle/
MPMATH = ["mp_rbfbrownian_n134"]


def rel(a, b, floor=1e-300):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


@pytest.mark.parametrize("name", SK + CLOSED + KALMAN + MPMATH)
def test_numpy_oracle_vs_golden(name):
    g = load_golden(name)
    f = go.fit(int(g["kernel_id"]), g["theta"], g["X"], g["y"])
    mu, var = go.predict(f, g["Xs"], include_noise=False)
    scale = max(float(np.max(np.abs(g["mean"]))), 1e-300)
    assert np.max(np.abs(mu - g["mean"])) / scale < 1e-9
    assert rel(var, g["var_latent"], 1e-12) < 1e-7
    assert abs(f.logml - float(g["logml"])) <= 1e-10 * abs(float(g["logml"]))
    if "alpha" in g:
        assert np.max(np.abs(f.alpha - g["alpha"])) / np.max(np.abs(g["alpha"])) < 1e-7


@pytest.mark.parametrize("name", SK + CLOSED + KALMAN + MPMATH)
def test_c_oracle_vs_golden(name, oracle_c):
    g = load_golden(name)
    rc, mu, var, logml, alpha, jit = oracle_c(int(g["kernel_id"]), g["theta"], g["X"], g["y"], g["Xs"], False)
    assert rc == 0 and jit == 0.0
    scale = max(float(np.max(np.abs(g["mean"]))), 1e-300)
    assert np.max(np.abs(mu - g["mean"])) / scale < 1e-9
    assert rel(var, g["var_latent"], 1e-12) < 1e-7
    assert abs(logml - float(g["logml"])) <= 1e-10 * abs(float(g["logml"]))


def test_gradient_vs_the_50_digit_pin():
    """The objective m.optimize() follows (gp_slip_node.py:36) and its gradient wrt (sigma_r^2, ell, sigma_b^2, sigma_n^2) at
    a working length-scale, against the 50-digit LU: the restatement's dL/dK = (alpha alpha^T - Ky^-1) / 2 contraction."""
    g = load_golden("mp_rbfbrownian_n134")
    nll, grad = go.nll_and_grad(2, g["theta"], g["X"], g["y"])
    assert abs(-nll - float(g["logml"])) <= 1e-12 * abs(float(g["logml"]))
    assert np.max(np.abs(-grad - g["dlogml_dtheta"])) <= 1e-10 * np.max(np.abs(g["dlogml_dtheta"]))


@pytest.mark.parametrize("name", SK[:5] + CLOSED + KALMAN + MPMATH)
def test_c_lapack_oracle_vs_golden(name, oracle_c_lapack):
    """oracle/gp_oracle_lapack.c (the single-thread LAPACK row of bench.py's cpu_baseline) against the same fixtures."""
    g = load_golden(name)
    rc, mu, var, logml, alpha, jit = oracle_c_lapack(int(g["kernel_id"]), g["theta"], g["X"], g["y"], g["Xs"], False)
    assert rc == 0 and jit == 0.0
    scale = max(float(np.max(np.abs(g["mean"]))), 1e-300)
    assert np.max(np.abs(mu - g["mean"])) / scale < 1e-9
    assert rel(var, g["var_latent"], 1e-12) < 1e-7
    assert abs(logml - float(g["logml"])) <= 1e-10 * abs(float(g["logml"]))


@pytest.mark.parametrize("name", ["slipval_window_rbfbrownian", "synth_window_rbfbrownian"])
def test_slip_node_callback_golden(name, oracle_c):
    g = load_golden(name)
    mean, sigma = go.slip_node_callback(g["time_array"], g["slip_array"], g["theta"])
    assert mean.shape == (599,) and sigma.shape == (599,)     # gap-free window -> 599
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(sigma, g["sigma"], rtol=1e-12)
    # the C restatement reproduces the node output through the same split/grid/slice
    X, Y, xtr, ytr = go.slip_node_split(g["time_array"], g["slip_array"])
    grid = go.slip_node_grid(X)[len(X):]
    rc, mu, var, _, _, _ = oracle_c(2, g["theta"], xtr, ytr[:, 0], grid[:, None], True)
    assert rc == 0
    assert np.max(np.abs(mu - g["mean"])) / np.max(np.abs(g["mean"])) < 1e-8
    np.testing.assert_allclose(2 * np.sqrt(var), g["sigma"], rtol=1e-8)


def test_split_and_grid_semantics():
    t = np.arange(11, 160, dtype=float)            # 149 ticks
    X, Y, xtr, ytr = go.slip_node_split(t, np.zeros_like(t))
    assert len(xtr) == int(0.9 * 149) == 134
    grid = go.slip_node_grid(X)
    assert grid[0] == 11 and grid[-1] == 159 + 599 and len(grid) == 149 + 599
    assert grid[len(X)] == 160                      # index slicing == first tick after the window
    # a window with a gap is still INDEX-sliced (gp_slip_node.py:59-61), so output starts earlier in time
    tg = np.concatenate([np.arange(11, 100), np.arange(110, 170)]).astype(float)
    Xg = tg.reshape(-1, 1)
    gg = go.slip_node_grid(Xg)
    assert gg[len(tg)] == 11 + len(tg) < tg.max() + 1


def test_properties_random():
    rng = np.random.default_rng(5)
    for kid, d in [(0, 2), (1, 4), (2, 1)]:
        N, M = 60, 17
        if kid == 2:
            X = np.sort(rng.uniform(5, 300, (N, 1)), 0)
            Xs = rng.uniform(5, 900, (M, 1))
            theta = np.array([0.7, 20.0, 0.02, 0.01])
        else:
            X, Xs = rng.normal(size=(N, d)), rng.normal(size=(M, d))
            theta = np.concatenate([[1.1], rng.uniform(0.5, 2, 1 if kid == 0 else d), [0.05]])
        y = rng.normal(size=N)
        f = go.fit(kid, theta, X, y, want_inverse=True)
        Ky = go.kernel_K(kid, theta, X) + (theta[-1] + go.GPY_DIAG_EPS) * np.eye(N)
        assert np.allclose(f.L @ f.L.T, Ky, rtol=1e-12, atol=1e-12)
        assert np.allclose(Ky @ f.alpha, y, rtol=1e-9, atol=1e-9)
        mu, var = go.predict(f, Xs)
        mu2, var2 = go.predict(f, Xs, via_inverse=True)      # literal GPy form
        assert np.allclose(mu, mu2) and np.allclose(var, var2, rtol=1e-8)
        assert np.all(var >= theta[-1])
        perm = rng.permutation(N)
        fp = go.fit(kid, theta, X[perm], y[perm])
        mup, varp = go.predict(fp, Xs)
        assert np.allclose(mu, mup, rtol=1e-8, atol=1e-10) and np.allclose(var, varp, rtol=1e-8)
        assert abs(f.logml - fp.logml) < 1e-8 * abs(f.logml)


def test_limits():
    # noise -> inf : mu -> 0, var -> k** + noise
    X = np.linspace(0, 1, 9)[:, None]
    y = np.sin(3 * X[:, 0])
    th = np.array([1.0, 0.5, 1e12])
    f = go.fit(0, th, X, y)
    mu, var = go.predict(f, np.array([[0.3]]))
    assert abs(mu[0]) < 1e-10 and abs(var[0] - (1.0 + 1e12)) / 1e12 < 1e-12
    # Brownian prior variance grows linearly with |x*| far from the data
    th = np.array([0.8, 2.0, 0.05, 0.01])
    f = go.fit(2, th, np.array([[10.0], [11.0], [12.0]]), np.array([0.1, 0.0, -0.1]))
    _, v = go.predict(f, np.array([[500.0], [1000.0]]), include_noise=False)
    assert np.allclose(v, 0.8 * 0.05 * np.array([500.0, 1000.0]), rtol=1e-9)


def test_jitchol_policy():
    A = np.array([[1.0, 1.0 + 1e-9], [1.0 + 1e-9, 1.0]])       # indefinite by 1e-9
    L, jitter, tries = go.jitchol(A)
    assert tries >= 1 and jitter == pytest.approx(1e-6 * 10 ** (tries - 1))
    with pytest.raises(np.linalg.LinAlgError):
        go.jitchol(np.array([[1.0, 2.0], [2.0, -1.0]]))


def test_c_oracle_jitter_and_failure(oracle_c):
    # duplicate inputs + zero noise: singular up to the 1e-8 GPy epsilon -> still factorises
    X = np.array([[0.0], [0.0], [1.0]])
    rc, mu, var, logml, alpha, jit = oracle_c(0, [1.0, 1.0, 0.0], X, np.array([1.0, 1.0, 0.0]), np.array([[0.5]]))
    assert rc == 0 and math.isfinite(logml)
    rc, *_ = oracle_c(0, [1.0, 1.0, -2.0], X, np.array([1.0, 1.0, 0.0]), np.array([[0.5]]))
    assert rc > 0                                               # negative noise: not PD, info returned


def test_lookahead_golden():
    g = load_golden("lookahead_restated")
    H = go.unpack_H(g["HvecData"], True)
    np.testing.assert_array_equal(H, g["H_client"])
    assert not np.array_equal(H, g["H_true"])                  # the r*4+c quirk loses information
    fired, cmd, i, xy = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], H,
                                        g["PosData"], float(g["arrival_time"]), float(g["now"]))
    assert fired == bool(g["fired"]) and i == int(g["i"])
    assert cmd == pytest.approx(float(g["stop_cmd"]), rel=1e-12)
    assert xy == pytest.approx(float(g["xy_err"]), rel=1e-10)
    # late result -> immediate 0.5 s stop (gp_predictor.cpp:107-111)
    fired, cmd, *_ = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], H,
                                     g["PosData"], 0.0, 1e6)
    assert fired and cmd == 0.5


def test_lookahead_oracle_against_the_50_digit_pin():
    """The restatement of GpPredictor::GPCallBack's loop pinned independently: mp_lookahead.npz is the C++ source
    (gp_predictor.cpp:64-99,144-178) restated in 50-digit arithmetic; the whole xy_err trace up to the 3.0 m threshold,
    the crossing step, i and the stop command must agree (fp64 loses ~1e-9 in the ECEF difference)."""
    g = load_golden("mp_lookahead")
    H = go.unpack_H(g["HvecData"], True)
    fired, cmd, i, xy, trace = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], H,
                                               g["PosData"], float(g["arrival_time"]), float(g["now"]), return_trace=True)
    assert fired and len(trace) == len(g["trace"]) and i == int(g["i_at"][-1])
    np.testing.assert_allclose(trace, g["trace"], rtol=1e-7)
    assert cmd == pytest.approx(float(g["stop_cmd"][-1]), rel=1e-12)
    for th, step, i_at in zip(g["thresholds"], g["cross_step"], g["i_at"]):
        f2, c2, i2, xy2, tr2 = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], H,
                                               g["PosData"], float(g["arrival_time"]), float(g["now"]), threshold=float(th),
                                               return_trace=True)
        assert f2 and len(tr2) == int(step) + 1 and i2 == int(i_at)


def test_llh_to_enu():
    g = load_golden("llh_to_enu_restated")
    np.testing.assert_allclose(go.llh_to_enu(*g["llh"]), g["enu"], rtol=1e-12, atol=1e-9)
    e = go.llh_to_enu(*go.INIT_LLH)
    assert np.linalg.norm(e) < 5.0       # the YAML origin LLH and ECEF agree to a few metres
    up = go.llh_to_enu(go.INIT_LLH[0], go.INIT_LLH[1], go.INIT_LLH[2] + 10.0)
    assert abs((up - e)[2] - 10.0) < 1e-6


@pytest.mark.parametrize("kid,d", [(0, 2), (1, 3), (2, 1)])
def test_nll_gradient_vs_finite_differences(kid, d):
    """Pins the analytic gradient of the restatement (GPy's dL/dK form) against central differences."""
    rng = np.random.default_rng(11 + kid)
    N = 40
    if kid == 2:
        X = (11.0 + np.arange(N))[:, None]
        theta = np.array([0.7, 12.0, 0.03, 0.02])
    else:
        X = rng.normal(size=(N, d))
        theta = np.concatenate([[0.9], rng.uniform(0.6, 1.8, 1 if kid == 0 else d), [0.08]])
    y = 0.2 * np.sin(np.arange(N) / 5.0) + 0.05 * rng.normal(size=N)
    nll, g = go.nll_and_grad(kid, theta, X, y)
    for p in range(len(theta)):
        h = 1e-6 * theta[p]
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h
        tm[p] -= h
        fd = (go.nll_and_grad(kid, tp, X, y)[0] - go.nll_and_grad(kid, tm, X, y)[0]) / (2 * h)
        assert g[p] == pytest.approx(fd, rel=2e-5, abs=1e-7)


def test_optimize_improves_likelihood():
    g = load_golden("synth_window_rbfbrownian")
    X, Y, xtr, ytr = go.slip_node_split(g["time_array"], g["slip_array"])
    th, lml, nev = go.optimize(2, xtr, ytr[:, 0])
    start = -go.nll_and_grad(2, np.ones(4), xtr, ytr[:, 0])[0]
    assert lml > start and np.all(th > 0) and nev <= 1000
    # stationarity in the transformed space
    nll, gr = go.nll_and_grad(2, th, xtr, ytr[:, 0])
    assert np.max(np.abs(gr * -np.expm1(-th))) < 1e-2


def test_fixture_classes():
    """tests/golden/README.md: every fixture says where its expected values come from; the ones the oracle itself produced
    (`restated`) are regression vectors, not pins, and each has a pin of the same quantity next to it."""
    import glob
    pins = {"closed", "kalman", "sklearn", "mpmath", "restated+scipy"}
    readme = open(os.path.join(GOLDEN, "README.md")).read()
    regression = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        name = os.path.splitext(os.path.basename(f))[0]
        src = str(load_golden(name)["source"])
        assert src in pins | {"restated"}, (name, src)
        if src == "restated":
            regression.append(name)
            assert name in readme.split("**regression**")[1], name
    assert sorted(regression) == ["llh_to_enu_restated", "lookahead_restated", "slipval_window_rbfbrownian", "synth_window_rbfbrownian"]
    for pin in ("mp_rbfbrownian_n134", "mp_lookahead"):
        assert str(load_golden(pin)["source"]) == "mpmath"
