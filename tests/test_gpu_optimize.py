"""GPU tests of the hyper-parameter path (the reference's `m.optimize()`,
gp_slip_node.py:36): the device gradient of the negative log marginal likelihood against the oracle
(which is itself pinned against finite differences on the CPU), and the optimiser against scipy's
L-BFGS-B on the oracle objective from the same start."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import gp_oracle as go
import corenav_gp_amd.synth as synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import corenav_gp_amd.engine as e
    e.load()
    return e


@pytest.mark.parametrize("kid,N,d", [(2, 134, 1), (2, 15, 1), (0, 200, 3), (1, 256, 6), (1, 300, 2), (1, 640, 6)])
def test_nll_grad_matches_oracle(engine, kid, N, d):
    rng = np.random.default_rng(7 * N + d)
    if kid == 2:
        X = (11.0 + np.arange(N))[:, None]
        theta = np.array([0.7, 12.0, 0.03, 0.02])
    else:
        X = rng.normal(size=(N, d))
        theta = np.concatenate([[0.9], rng.uniform(0.6, 1.8, 1 if kid == 0 else d), [0.08]])
    y = 0.2 * np.sin(np.arange(N) / 5.0) + 0.05 * rng.normal(size=N)
    ctx = engine.Context(max_n=N, max_m=N, max_d=d)
    nll, g = ctx.nll_grad(X, y, kid, theta)
    onll, og = go.nll_and_grad(kid, theta, X, y)
    assert abs(nll - onll) <= 1e-6 * abs(onll)
    # 1e-6 relative to the gradient's scale (single components can pass through zero)
    assert np.max(np.abs(g - og)) <= 1e-6 * np.max(np.abs(og))
    # the context is left fitted at theta: prediction still works after a gradient evaluation
    Xs = X[:5] + 0.25
    mu, var = ctx.predict(Xs)
    f = go.fit(kid, theta, X, y)
    omu, ovar = go.predict(f, Xs)
    assert np.max(np.abs(mu - omu)) <= 1e-6 * np.max(np.abs(omu)) and np.max(np.abs(var - ovar) / ovar) < 1e-6


def _check_8c(g, lml, mean, sigma):
    """SURVEY 8c, optimiser row: logML(theta_hat) >= logML(fixture) - 1e-6 |logML|, published mean / sigma within 1e-3."""
    assert lml >= float(g["logml"]) - 1e-6 * abs(float(g["logml"]))
    assert mean.shape == g["mean"].shape
    assert np.max(np.abs(mean - g["mean"])) <= 1e-3 * np.max(np.abs(g["mean"]))
    assert np.max(np.abs(sigma - g["sigma"]) / g["sigma"]) <= 1e-3


@pytest.mark.parametrize("name", ["slipval_window_opt", "synth_window_opt"])
def test_optimize_reference_window(engine, name):
    """The reference node's flow (gp_slip_node.py:31-36,45-61): all-ones start (GPy defaults), optimise, publish 599 points
    -- against the committed optimised-theta fixture (scipy's L-BFGS-B on the oracle objective) at SURVEY 8c's bars, through
    cgp_optimize, through the fused callback cgp_slip_node_callback_opt, and through cgp_optimize_batch +
    cgp_fit_predict_batch."""
    g = load_golden(name)
    t, s = g["time_array"], g["slip_array"]
    X, Y, xtr, ytr = go.slip_node_split(t, s)
    ctx = engine.Context(max_n=256, max_m=1024, max_d=1)
    th, lml, nev = ctx.optimize(xtr, ytr[:, 0], 2, g["theta0"])
    assert nev <= 1000 and np.all(th > 0)
    # the value reported is the true logML at the returned theta
    assert lml == pytest.approx(-go.nll_and_grad(2, th, xtr, ytr[:, 0])[0], rel=1e-8)
    # the optimiser is scipy's algorithm (lbfgs_core.hpp): same trajectory length as the fixture's run, same optimum
    assert abs(nev - int(g["n_evals"])) <= 2
    np.testing.assert_allclose(th, g["theta"], rtol=1e-4)
    mean, sigma, th2 = ctx.slip_node_callback_opt(t, s, g["theta0"])
    _check_8c(g, -go.nll_and_grad(2, th2, xtr, ytr[:, 0])[0], mean, sigma)
    _check_8c(g, lml, mean, sigma)
    # what was published is the oracle's answer at the theta the engine returned (fixed-theta bar)
    em, es = go.slip_node_callback(t, s, th2)
    assert np.max(np.abs(mean - em)) <= 1e-6 * np.max(np.abs(em)) and np.max(np.abs(sigma - es) / es) < 1e-6
    # batched entry points: two copies of the window
    n = len(xtr)
    bctx = engine.Context(max_n=n, max_m=1024, max_d=1, max_batch=2)
    Xb, yb = np.stack([xtr, xtr]), np.stack([ytr[:, 0], ytr[:, 0]])
    thb, lmlb, nevb = bctx.optimize_batch(Xb, yb, 2, g["theta0"])
    grid = go.slip_node_grid(X)[len(X):]
    rc, mb, vb, lb, info = bctx.fit_predict_batch(Xb, yb, np.stack([grid, grid])[:, :, None], thb, 2)
    assert rc == 0 and not info.any()
    for b in range(2):
        _check_8c(g, lmlb[b], mb[b], 2.0 * np.sqrt(vb[b]))


def test_optimize_se_ard(engine):
    """A window of the large-window machinery (N = 192 > 160: host L-BFGS over device gradients) against its fixture."""
    g = load_golden("synth_se_ard_n192_d3_opt")
    ctx = engine.Context(max_n=192, max_m=192, max_d=3)
    th, lml, nev = ctx.optimize(g["X"], g["y"], 1, g["theta0"])
    assert lml >= float(g["logml"]) - 1e-6 * abs(float(g["logml"]))
    assert lml > -go.nll_and_grad(1, g["theta0"], g["X"], g["y"])[0]
    assert abs(nev - 1 - int(g["n_evals"])) <= 3          # n_evals counts the final refit at the optimum on this path
    mu, var = ctx.predict(g["Xs"])
    assert np.max(np.abs(mu - g["mean"])) <= 1e-3 * np.max(np.abs(g["mean"]))
    assert np.max(np.abs(var - g["var"]) / g["var"]) <= 1e-3


def test_node_mirror_optimises_like_the_reference(engine):
    """GpSlipNode default = the reference behaviour: GPy start values, optimise, publish 599 points."""
    from corenav_gp_amd import gp_slip_node as node
    g = load_golden("synth_window_rbfbrownian")
    n = node.GpSlipNode()
    out = n.callback(node.GP_Input(g["time_array"], g["slip_array"]))
    assert len(out.mean) == 599 and np.all(np.isfinite(out.sigma)) and np.all(out.sigma > 0)
    X, Y, xtr, ytr = go.slip_node_split(g["time_array"], g["slip_array"])
    assert -go.nll_and_grad(2, n.last_theta, xtr, ytr[:, 0])[0] > -go.nll_and_grad(2, np.ones(4), xtr, ytr[:, 0])[0]


def test_optimize_batch_matches_single(engine):
    """Batched optimiser: every window follows the same L-BFGS trajectory as the single-window call
    (same objective values per evaluation, so the same optimum up to rounding)."""
    B, N = 6, 134
    Xs, ys = [], []
    for b in range(B):
        t, s = synth.reference_window(149, tick0=11 + 7 * b, seed=synth.SEED_BASE + 50 + b)
        Xs.append(t[:N, None])
        ys.append(s[:N])
    X, y = np.stack(Xs), np.stack(ys)
    ctx = engine.Context(max_n=N, max_m=N, max_d=1, max_batch=B)
    th, lml, nev = ctx.optimize_batch(X, y, 2, np.ones(4))
    ctx1 = engine.Context(max_n=N, max_m=N, max_d=1)
    for b in range(B):
        th1, lml1, nev1 = ctx1.optimize(X[b], y[b], 2, np.ones(4))
        assert lml[b] == pytest.approx(lml1, rel=1e-9) and nev[b] <= 1000
        np.testing.assert_allclose(th[b], th1, rtol=1e-6)
        olml = go.optimize(2, X[b], y[b])[1]
        assert lml[b] >= olml - 1e-6 * abs(olml)


def test_optimize_batch_climbs_the_jitter_ladder(engine):
    """GPy's jitchol inside m.optimize(): a trial point whose matrix is not positive definite gets
    mean(diag) 1e-6 10^k.  Window 1 has duplicated inputs and starts at amplitude 1e9 with (almost) no
    noise, so its first factorisations fail without jitter; the batched optimiser must follow the same
    trajectory as the single-window one (which always had the ladder) instead of treating the point as
    infeasible, and the healthy window next to it must be unaffected."""
    N = 96
    rng = np.random.default_rng(77)
    xa = np.sort(rng.normal(size=N))
    xb = np.repeat(np.sort(rng.normal(size=N // 2)), 2)
    X = np.stack([xa, xb])[:, :, None]
    y = np.sin(2.0 * X[:, :, 0]) + 0.01 * rng.normal(size=(2, N))
    y[1] = np.repeat(y[1][::2], 2)
    th0 = np.array([[1.0, 1.0, 1.0], [1e9, 1.0, 1e-10]])
    ctx = engine.Context(max_n=N, max_m=N, max_d=1, max_batch=2)
    th, lml, nev = ctx.optimize_batch(X, y, 0, th0.copy(), max_evals=60)
    ctx1 = engine.Context(max_n=N, max_m=N, max_d=1)
    assert ctx1.nll_grad(X[1], y[1], 0, th0[1])[0] is not None and ctx1.last_jitter() > 0   # the ladder is needed
    for b in range(2):
        th1, lml1, nev1 = ctx1.optimize(X[b], y[b], 0, th0[b].copy(), max_evals=60)
        assert np.isfinite(lml[b]) and lml[b] == pytest.approx(lml1, rel=1e-6)
        np.testing.assert_allclose(th[b], th1, rtol=1e-4)


@pytest.mark.parametrize("B", [14, 32])
def test_gradient_mode_on_the_mid_size_schedule(engine, B):
    """A batched gradient evaluation (gradient mode: the M = N "test rows" are the identity) of 14 / 32 fp64 windows of three
    block steps takes the mid-size throughput schedule -- kind C pre-update, and from 28 fits the extra rows on a second
    stream: value and gradient of every window equal the single-window call (latency schedule) to rounding."""
    N, d = 300, 3
    rng = np.random.default_rng(B)
    X = rng.normal(size=(B, N, d))
    y = 0.2 * np.sin(np.arange(N) / 5.0)[None] + 0.05 * rng.normal(size=(B, N))
    th = np.stack([np.concatenate([[0.9], rng.uniform(0.6, 1.8, d), [0.08]]) for _ in range(B)])
    ctx = engine.Context(max_n=N, max_m=N, max_d=d, max_batch=B)
    th_b, lml_b, nev = ctx.optimize_batch(X, y, 1, th, max_evals=1)     # one evaluation: logML at the start point
    one = engine.Context(max_n=N, max_m=N, max_d=d)
    for b in (0, B // 2, B - 1):
        nll, g = one.nll_grad(X[b], y[b], 1, th[b])
        onll, og = go.nll_and_grad(1, th[b], X[b], y[b])
        assert abs(nll - onll) <= 1e-6 * abs(onll) and np.max(np.abs(g - og)) <= 1e-6 * np.max(np.abs(og))
    # the batched optimiser's own first evaluations, through a few real rounds
    th2, lml2, nev2 = ctx.optimize_batch(X, y, 1, th, max_evals=6)
    for b in (0, B - 1):
        t1, l1, n1 = one.optimize(X[b], y[b], 1, th[b], max_evals=6)
        assert lml2[b] == pytest.approx(l1, rel=1e-8) and np.allclose(th2[b], t1, rtol=1e-6)


def test_short_window_kernel_matches_the_large_window_machinery(engine):
    """Windows of at most 160 samples take the one-launch kernel (csrc/cgp_small.hpp: Gram, Cholesky, inverse, gradient
    sums, jitter ladder and the L-BFGS loop inside one workgroup); CGP_SMALL=off keeps the large-window machinery for them
    (seven launches per evaluation, L-BFGS on the host).  Same objective, same optimiser, different summation orders: the
    two must agree far inside the parity bar on an evaluation and to the optimiser's own tolerance on the optimum."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import json, numpy as np, torch\n"
        "import corenav_gp_amd.engine as e\n"
        "from tests.conftest import load_golden\n"
        "from oracle import gp_oracle as go\n"
        "g = load_golden('slipval_window_rbfbrownian')\n"
        "X, Y, xtr, ytr = go.slip_node_split(g['time_array'], g['slip_array'])\n"
        "ctx = e.Context(max_n=256, max_m=1024, max_d=1)\n"
        "nll, gr = ctx.nll_grad(xtr, ytr[:, 0], 2, np.array([0.7, 12.0, 0.03, 0.02]))\n"
        "th, lml, nev = ctx.optimize(xtr, ytr[:, 0], 2, np.ones(4))\n"
        "m, s, th2 = ctx.slip_node_callback_opt(g['time_array'], g['slip_array'], np.ones(4))\n"
        "print(json.dumps({'nll': nll, 'grad': gr.tolist(), 'theta': th.tolist(), 'lml': lml, 'nev': nev, 'mean': m.tolist(),"
        " 'sigma': s.tolist(), 'theta2': th2.tolist()}))\n")
    out = {}
    for mode in ("on", "off"):
        env = dict(os.environ, CGP_SMALL=mode, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = out["on"], out["off"]
    assert a["nll"] == pytest.approx(b["nll"], rel=1e-10)
    np.testing.assert_allclose(a["grad"], b["grad"], rtol=1e-7, atol=1e-9 * np.max(np.abs(b["grad"])))
    # the optimum: both stop on the same criteria; theta agrees to the optimiser's resolution, logML much tighter
    assert a["lml"] == pytest.approx(b["lml"], rel=1e-7)
    np.testing.assert_allclose(a["theta"], b["theta"], rtol=2e-3)
    np.testing.assert_allclose(a["theta2"], a["theta"], rtol=1e-12)      # the fused callback runs the same optimisation
    np.testing.assert_allclose(a["mean"], b["mean"], rtol=0, atol=2e-3 * np.max(np.abs(b["mean"])))
    np.testing.assert_allclose(a["sigma"], b["sigma"], rtol=2e-3)
    assert abs(a["nev"] - b["nev"]) <= 6


def test_tick_grid_table_is_bitwise_the_direct_evaluation():
    """RBF x Brownian windows whose inputs are tick counts (integer-valued, as the reference's GP_Input always is) take the RBF
    factor from a per-evaluation table in LDS (csrc/cgp_small.hpp: SmallArgs::tab_n).  r^2 is then an exact integer, the table
    entry is the same expression, and every output must carry the same BITS as with the table switched off (CGP_TICKTAB=off,
    read once per process: child processes) -- gradient, optimum, callback mean and sigma; gaps in the ticks, a large tick
    offset; the batched entry points (cgp_fit_predict_batch, cgp_optimize_batch); and a window off the grid (x + 0.25) must simply
    take the direct path."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from corenav_gp_amd import engine\n"
        "g = np.load(%r)\n"
        "t, s, th = g['time_array'], g['slip_array'], g['theta']\n"
        "out = {}\n"
        "for name, tt in (('plain', t), ('gaps', np.delete(t + 40000.0, [5, 6, 40, 90]) + np.r_[np.zeros(60), 7 * np.ones(len(t) - 64)]), ('offgrid', t + 0.25)):\n"
        "    ss = s[:len(tt)]\n"
        "    ctx = engine.Context(max_n=256, max_m=1024, max_d=1)\n"
        "    n = int(0.9 * len(tt))\n"
        "    nll, gr = ctx.nll_grad(tt[:n, None], ss[:n], engine.KERNEL_RBF_BROWNIAN, th)\n"
        "    m, sg = ctx.slip_node_callback(tt, ss, th)\n"
        "    m2, sg2, tho = ctx.slip_node_callback_opt(tt, ss, np.ones(4))\n"
        "    out[name] = [float(nll).hex()] + [float(v).hex() for v in np.concatenate([gr, m, sg, m2, sg2, tho])]\n"
        "n, W = int(0.9 * len(t)), 5\n"
        "X = np.stack([t[:n] + 1000.0 * k for k in range(W)])[:, :, None]; y = np.stack([np.roll(s, k)[:n] for k in range(W)])\n"
        "Xs = np.stack([X[k, -1, 0] + 1 + np.arange(77.0) for k in range(W)])[:, :, None]\n"
        "bctx = engine.Context(max_n=n, max_m=256, max_d=1, max_batch=W)\n"
        "rc, mean, var, logml, info = bctx.fit_predict_batch(X, y, Xs, np.tile(th, (W, 1)), engine.KERNEL_RBF_BROWNIAN)\n"
        "tb, lb, eb = bctx.optimize_batch(X, y, engine.KERNEL_RBF_BROWNIAN, np.ones(4), max_evals=30)\n"
        "assert rc == 0 and not info.any()\n"
        "out['batch'] = [float(v).hex() for v in np.concatenate([mean.ravel(), var.ravel(), logml, tb.ravel(), lb])]\n"
        "print('RESULT' + json.dumps(out))\n" % (root, os.path.join(root, "tests", "golden", "slipval_window_rbfbrownian.npz")))

    def run(env):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][6:])

    on, off = run({}), run({"CGP_TICKTAB": "off"})
    for name in ("plain", "gaps", "offgrid", "batch"):
        assert on[name] == off[name], name
