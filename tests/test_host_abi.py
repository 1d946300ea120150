"""CPU tests: the C-ABI library loads and exports every symbol include/corenav_gp.h declares, and
the host-side (non-GPU) entry points -- GpPredictor look-ahead, llh_to_enu, argument checking --
match the oracle.  No GPU compute is attempted here."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import gp_oracle as go

engine = pytest.importorskip("corenav_gp_amd.engine")
if not os.path.exists(engine.LIB_PATH):
    import __graft_entry__ as ge
    ge.build()


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "corenav_gp.h")).read()
    declared = set(re.findall(r"\b(cgp_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"cgp_ctx"}
    lib = engine.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in corenav_gp.h but not exported"
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    assert lib.cgp_abi_version() == engine.ABI_VERSION == 3


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.Context()


def test_create_ex_tells_why_it_failed():
    """cgp_create returns a bare NULL; cgp_create_ex distinguishes an argument out of range from a missing / foreign device
    (and, on a GPU box, from out-of-memory)."""
    import ctypes
    lib = engine.load()
    st = ctypes.c_int(12345)
    assert not lib.cgp_create_ex(0, 0, 8, 1, 1, engine.F64, ctypes.byref(st)) and st.value == -1            # max_n < 1: CGP_EINVAL
    assert not lib.cgp_create_ex(0, 64, 8, engine.MAX_D + 1, 1, engine.F64, ctypes.byref(st)) and st.value == -1
    assert not lib.cgp_create_ex(0, 64, 8, 1, 1, 7, ctypes.byref(st)) and st.value == -1                    # unknown dtype
    assert not lib.cgp_create_ex(9999, 64, 8, 1, 1, engine.F64, ctypes.byref(st))
    assert lib.cgp_strerror(st.value) == b"no usable gfx950 device"
    assert not lib.cgp_create_ex(9999, 64, 8, 1, 1, engine.F64, None)                                        # status may be NULL


def test_strerror():
    lib = engine.load()
    assert lib.cgp_strerror(0) == b"ok"
    assert b"positive definite" in lib.cgp_strerror(17)
    assert b"invalid" in lib.cgp_strerror(-1)


def test_llh_to_enu_matches_oracle():
    g = load_golden("llh_to_enu_restated")
    np.testing.assert_allclose(engine.llh_to_enu(*g["llh"]), g["enu"], rtol=1e-12, atol=1e-9)
    rng = np.random.default_rng(3)
    for _ in range(20):
        llh = np.array(go.INIT_LLH) + rng.normal(0, [1e-5, 1e-5, 30.0])
        np.testing.assert_allclose(engine.llh_to_enu(*llh), go.llh_to_enu(*llh), rtol=1e-11, atol=1e-8)


@pytest.mark.parametrize("bug", [True, False])
def test_predict_stop_matches_oracle(bug):
    g = load_golden("lookahead_restated")
    H = go.unpack_H(g["HvecData"], bug) if bug else g["H_true"]
    hvec = g["HvecData"] if bug else g["H_true"].reshape(60)
    exp = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], H, g["PosData"],
                          float(g["arrival_time"]), float(g["now"]))
    got = engine.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], hvec,
                              g["PosData"], float(g["arrival_time"]), float(g["now"]), h_bug_compatible=bug)
    assert got[0] == exp[0] and got[2] == exp[2]
    assert got[1] == pytest.approx(exp[1], rel=1e-12)
    assert got[3] == pytest.approx(exp[3], rel=1e-9)
    if bug:
        assert got[2] == int(g["i"]) and got[1] == pytest.approx(float(g["stop_cmd"]), rel=1e-12)


def test_predict_stop_against_the_50_digit_pin():
    """Rows a10 / a11 pinned independently of oracle/: tests/golden/mp_lookahead.npz is the reference's loop
    (gp_predictor.cpp:64-99) and llh_to_enu (:144-178) restated from the C++ source in 50-digit arithmetic.  A ladder of
    thresholds samples the xy_err trace: the step that crosses each, the odometry index i, the published stop command
    (:107-118) and the error at the crossing must be the pin's."""
    g = load_golden("mp_lookahead")
    for th, i_at, xy_at, cmd_at in zip(g["thresholds"], g["i_at"], g["xy_at"], g["stop_cmd"]):
        fired, cmd, i, xy = engine.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"],
                                                g["HvecData"], g["PosData"], float(g["arrival_time"]), float(g["now"]),
                                                threshold=float(th))
        assert fired and i == int(i_at)
        assert xy == pytest.approx(float(xy_at), rel=1e-6) and cmd == pytest.approx(float(cmd_at), rel=1e-12)


def test_predict_stop_edge_cases():
    g = load_golden("lookahead_restated")
    args = (g["PvecData"], g["QvecData"], g["STMvecData"], g["HvecData"], g["PosData"])
    # empty horizon: loop body never runs (gp_predictor.cpp:64), nothing published
    fired, cmd, i, xy = engine.predict_stop(np.zeros(0), np.zeros(0), *args)
    assert not fired and i == 0
    # late result -> immediate stop 0.5 s (gp_predictor.cpp:107-111)
    fired, cmd, i, xy = engine.predict_stop(g["mean"], g["sigma"], *args, arrival_time=0.0, now=1e6)
    assert fired and cmd == 0.5
    # huge threshold: never fires, consumes the whole horizon
    fired, cmd, i, xy = engine.predict_stop(g["mean"], g["sigma"], *args, threshold=1e9)
    assert not fired and i == len(g["mean"])
    exp = go.predict_stop(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"],
                          go.unpack_H(g["HvecData"]), g["PosData"], threshold=1e9)
    assert xy == pytest.approx(exp[3], rel=1e-8)


def test_synth_generator_is_deterministic():
    import corenav_gp_amd.synth as synth
    k1 = synth.config(2, batch=2, N=256)
    k2 = synth.config(2, batch=2, N=256)
    for a, b in zip(k1[1:5], k2[1:5]):
        np.testing.assert_array_equal(a, b)
    assert k1[1].shape == (2, 256, 6) and k1[3].shape == (2, 599, 6)
    t, s = synth.reference_window()
    assert len(t) == 149 and np.all(np.diff(t) == 1) and np.all(np.abs(s) < 1)


def test_gppredictor_class_callback():
    """The GpPredictor C++ class (reference API names) end to end on an in-process NodeHandle."""
    g = load_golden("lookahead_restated")
    npub, cmd = engine.gppredictor_callback(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"],
                                            g["HvecData"], g["PosData"], float(g["arrival_time"]), float(g["now"]))
    assert npub == 1 and cmd == pytest.approx(float(g["stop_cmd"]), rel=1e-12)
    # nothing is published when the threshold is never crossed (empty horizon)
    npub, cmd = engine.gppredictor_callback(np.zeros(0), np.zeros(0), g["PvecData"], g["QvecData"], g["STMvecData"],
                                            g["HvecData"], g["PosData"], 0.0, 0.0)
    assert npub == 0


@pytest.mark.parametrize("n", [2, 5, 10])
def test_lbfgs_selftest_rosenbrock(n):
    """The host L-BFGS behind cgp_optimize (csrc/lbfgs.hpp) on Rosenbrock, against scipy's L-BFGS-B."""
    import ctypes
    import scipy.optimize as so
    x = np.full(n, -1.2)
    x[1::2] = 1.0
    f = ctypes.c_double(0.0)
    dp = ctypes.POINTER(ctypes.c_double)
    nev = engine.load().cgp_selftest_lbfgs(x.ctypes.data_as(dp), n, 2000, ctypes.byref(f))
    assert 0 < nev <= 2000
    np.testing.assert_allclose(x, np.ones(n), atol=1e-5)
    assert f.value < 1e-10
    ref = so.minimize(so.rosen, np.where(np.arange(n) % 2, 1.0, -1.2), jac=so.rosen_der, method="L-BFGS-B")
    assert nev < 4 * ref.nfev + 50      # same order of work as the reference optimiser


def test_gppredictor_header_offers_the_reference_signature():
    """inside a catkin workspace (<ros/ros.h>, <Eigen/Dense> found) csrc/gp_predictor.h is the
    reference's class -- GpPredictor(ros::NodeHandle &), typedef Eigen::MatrixXd Matrix, the private
    gp_sub_ / stop_cmd_pub_ / clt_setStopping_ / nh_ (gp_predictor.h:22,25,57-60) -- and the node's main is
    the reference's main.  That configuration cannot be compiled in this image (no roscpp, no Eigen); this
    checks the surface is spelled out and that the ROS-free configuration, which the library ships, builds."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "corenav_gp_amd", "csrc")
    h = open(os.path.join(csrc, "gp_predictor.h")).read()
    for needle in ("typedef ros::NodeHandle NodeHandle;", "GpPredictor(corenav_types::NodeHandle &);",
                   "typedef Eigen::MatrixXd Matrix;", "ros::Subscriber gp_sub_;", "ros::Publisher stop_cmd_pub_;",
                   "ros::ServiceClient clt_setStopping_;", "corenav_types::NodeHandle &nh_;",
                   "typedef Eigen::Matrix<double, 3, 1> Vector3;", "int main(int argc, char **argv);"):
        assert needle in h, needle
    cpp = open(os.path.join(csrc, "gp_predictor.cpp")).read()
    for needle in ('nh.subscribe("/core_nav/core_nav/gp_result", 1, &GpPredictor::GPCallBack, this)',
                   'serviceClient<core_nav::SetStopping>("/core_nav/core_nav/stopping_service")',
                   'advertise<std_msgs::Float64>("/core_nav/core_nav/stop_cmd", 1)'):
        assert needle in cpp, needle
    node = open(os.path.join(ROOT, "corenav_gp_amd", "ros", "gp_predictor_node.cpp")).read()
    body = re.sub(r"\s+", " ", node[node.index("int main"):])
    assert 'ros::init(argc, argv, "gp_predictor"); ros::NodeHandle nh(""); GpPredictor gp_predictor(nh); ros::spin();' in body
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DCORENAV_NO_ROS", "-DCORENAV_NO_EIGEN",
                        os.path.join(csrc, "gp_predictor.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
