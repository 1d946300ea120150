"""CPU tests of the producer side: the C++ SlipWindowRecorder behind the C ABI against
the Python restatement of CoreNav::Update's slip + recording-window state machine
(CoreNav.cpp:176,244-330,755-759,794-816), on randomised drive / stop / restart streams."""
import numpy as np
import pytest

from oracle import gp_oracle as go

engine = pytest.importorskip("corenav_gp_amd.engine")


def drive_stream(seed, T=1500, stops=True):
    rng = np.random.default_rng(seed)
    ticks = []
    cmd = 1.0
    stop_until = -1
    for t in range(T):
        if stops and stop_until < 0 and rng.random() < 0.004:
            stop_until = t + int(rng.integers(20, 80))
        if t <= stop_until:
            cmd, v = 0.0, 0.0
        else:
            if stop_until >= 0 and t > stop_until:
                stop_until = -1
            cmd = 1.0
            v = 0.8
        vlin = v * (1.0 - (0.1 * np.sin(t / 40.0) + 0.05 * rng.normal())) if v > 0 else 0.0
        wheels = [v * (1 + 0.01 * rng.normal()) if v > 0 else 0.0 for _ in range(4)]
        ticks.append((wheels, vlin, cmd))
    return ticks


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_recorder_matches_oracle(seed):
    rec, orc = engine.SlipRecorder(), go.SlipRecorderOracle()
    rng = np.random.default_rng(100 + seed)
    nwin = 0
    for wheels, vlin, cmd in drive_stream(seed):
        rec.cmd_callback(cmd)
        orc.cmd_callback(cmd)
        got = rec.update(*wheels, vlin, cmd)
        exp = orc.update(*wheels, vlin, cmd)
        assert (got is None) == (exp is None)
        assert rec.slip == orc.slip or (np.isnan(rec.slip) and np.isnan(orc.slip))
        if got is not None:
            nwin += 1
            np.testing.assert_array_equal(got[0], exp[0])
            np.testing.assert_array_equal(got[1], exp[1])
            assert 15 <= len(got[0]) <= 149                      # CoreNav.cpp:278,300
            stop = float(rng.uniform(0.5, 25.0))                 # what gp_predictor would answer
            rec.stop_callback(stop)
            orc.stop_callback(stop)
        st = rec.state()
        assert st["odomUptCount"] == orc.odomUptCount and st["stopRecording"] == orc.stopRecording
        assert bool(st["gp_flag"]) == orc.gp_flag and bool(st["first_driving_flag"]) == orc.first_driving_flag
    assert nwin >= 2


def test_first_window_timing_and_size():
    rec = engine.SlipRecorder()
    out, when = None, None
    for t, (wheels, vlin, cmd) in enumerate(drive_stream(9, T=400, stops=False)):
        got = rec.update(*wheels, vlin, cmd)
        if got is not None and out is None:
            out, when = got, t + 1
    # driving from tick 1: start = 1 + 10, stop = 161; samples 12..160 -> 149, published at tick 161
    assert when == 161 and len(out[0]) == 149 and out[0][0] == 12 and out[0][-1] == 160
    assert np.all(np.diff(out[0]) == 1)


def test_short_window_is_skipped_and_slip_clamps():
    rec = engine.SlipRecorder()
    # stationary rover: |rearVel| < 0.001 -> slip forced to 0, nothing recorded
    assert rec.update(0.0, 0.0, 0.0, 0.0, 0.0, 1.0) is None and rec.slip == 0.0
    # INS much faster than the wheels: raw slip < -1 clamps to -1 and is excluded from windows
    assert rec.update(0.5, 0.5, 0.5, 0.5, 2.0, 1.0) is None and rec.slip == -1.0
    assert rec.state()["first_driving_flag"] == 1.0


def test_window_longer_than_the_callers_buffer_is_an_error():
    """cgp_recorder_update never truncates silently: a 149-sample window into a 100-entry buffer returns
    CGP_ECAPACITY with *n_out = 149."""
    rec = engine.SlipRecorder()
    seen = False
    for wheels, vlin, cmd in drive_stream(9, T=200, stops=False):
        try:
            got = rec.update(*wheels, vlin, cmd, cap=100)
            assert got is None
        except engine.CgpError as e:
            assert e.code == -6
            seen = True
            break
    assert seen
