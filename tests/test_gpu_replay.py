"""GPU test of the closed loop (stand-in for BASELINE configs[4]): synthetic rover ->
recorder (C++) -> GP engine (HIP) -> GpPredictor (C++) -> stop_cmd -> recorder / drive FSM, for a
small Monte-Carlo ensemble, with every published GP_Output checked against the oracle."""
import numpy as np
import pytest

from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def test_closed_loop_ensemble():
    from corenav_gp_amd import replay
    ens = replay.ClosedLoopEnsemble(n_traj=6)
    npub = ens.run(900)                    # 90 s of driving
    assert npub >= 6                       # every trajectory published at least its first window
    for tr in ens.traj:
        assert len(tr.windows) >= 1 and len(tr.windows) == len(tr.results)
        for (t, s), (mean, sigma, th) in zip(tr.windows, tr.results):
            em, es = go.slip_node_callback(t, s, th)
            assert mean.shape == em.shape
            assert np.max(np.abs(mean - em)) <= 1e-6 * np.max(np.abs(em))
            assert np.max(np.abs(sigma - es) / es) < 1e-6
        assert len(tr.stop_cmds) >= 1 and all(c > 0 for c in tr.stop_cmds)
        assert tr.stops >= 1               # the rover actually stopped (ZUPT opportunity)
        st = tr.rec.state()
        assert st["stopRecording"] > 161   # the next window was re-armed after the stop (CoreNav.cpp:311-321)
    # first window: ticks 12..160, published at tick 161 for every trajectory
    assert all(len(tr.windows[0][0]) == 149 for tr in ens.traj)


def test_closed_loop_with_optimiser():
    from corenav_gp_amd import replay
    ens = replay.ClosedLoopEnsemble(n_traj=2, optimize=True)
    ens.run(200)
    for tr in ens.traj:
        (t, s), (mean, sigma, th) = tr.windows[0], tr.results[0]
        X, Y, xtr, ytr = go.slip_node_split(t, s)
        assert -go.nll_and_grad(2, th, xtr, ytr[:, 0])[0] > -go.nll_and_grad(2, np.ones(4), xtr, ytr[:, 0])[0]
        em, es = go.slip_node_callback(t, s, th)
        assert np.max(np.abs(mean - em)) <= 1e-6 * np.max(np.abs(em))
