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


def test_replay_ensemble_two_ranks_match_one():
    """BASELINE configs[4] stand-in across ranks: tools/replay_ensemble.py with two ranks (both on GPU 0, summaries
    over gloo -- RCCL refuses two ranks on one device) gathers, in global trajectory order, exactly the
    per-trajectory summaries one rank computes for the whole ensemble: the block partition and the seeds
    do not depend on the world size."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "replay_ensemble.py")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}

    def line(out):
        return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    one = subprocess.run([sys.executable, tool, "--traj", "6", "--ticks", "400"], capture_output=True, text=True,
                         timeout=600, env=base)
    assert one.returncode == 0, one.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CGP_BENCH_SAME_DEVICE="1", CGP_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, tool, "--traj", "6", "--ticks", "400"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:] + outs[1][1][-2000:]
    a, b = line(one.stdout), line(outs[0][0])
    assert b["n_gpus"] == 2 and a["n_gpus"] == 1
    # (6 windows per call or 3 per rank: both take the latency schedule, whose results do not depend on the call
    # size; the stop times are compared to 1e-9 s all the same, the counts exactly)
    assert np.allclose(np.array(a["per_trajectory"]), np.array(b["per_trajectory"]), rtol=0, atol=1e-9)
    assert a["windows"] == b["windows"] >= 6 and a["stops"] == b["stops"]
