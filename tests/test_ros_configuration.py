"""the ROS configuration of OUR GpPredictor -- GpPredictor(ros::NodeHandle &), Eigen members, the
private gp_sub_ / stop_cmd_pub_ / clt_setStopping_ wired as gp_predictor.cpp:9-14, and the node's main -- is compiled
(against the API doubles of tests/ros_api_doubles: this image has no roscpp, no Eigen, no catkin) and driven: one
GP_Output delivered through the subscription the class registered, its SetStopping call answered from a canned filter
snapshot, and what it publishes compared with the reference-restated numbers and with the POD configuration of the
same class behind the C ABI.  What this does not show: a build against the real roscpp (ros/CMakeLists.txt)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "corenav_gp_amd", "csrc")
DOUBLES = os.path.join(ROOT, "tests", "ros_api_doubles")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("ros_cfg")
    exe = str(out / "drive_gp_predictor")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", DOUBLES, "-I", CSRC, "-o", exe,
           os.path.join(DOUBLES, "drive_gp_predictor.cpp"), os.path.join(CSRC, "gp_predictor.cpp"),
           os.path.join(CSRC, "gp_predictor_core.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the node's translation unit (reference main: gp_predictor.cpp:180-190) compiles in the same configuration
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", DOUBLES, "-I", CSRC, "-c", "-o", str(out / "node.o"),
                        os.path.join(ROOT, "corenav_gp_amd", "ros", "gp_predictor_node.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-C", str(out / "node.o")], capture_output=True, text=True).stdout
    assert " T main" in syms and "GpPredictor::GpPredictor(ros::NodeHandle&)" in syms
    return exe, out


def run(driver, g, mean, sigma, arrival, now):
    exe, out = driver
    path = str(out / "in.txt")
    with open(path, "w") as f:
        f.write(f"{len(mean)}\n")
        for a in (mean, sigma, g["PvecData"], g["QvecData"], g["STMvecData"], g["HvecData"], g["PosData"], [arrival, now]):
            f.write(" ".join(repr(float(x)) for x in np.asarray(a).reshape(-1)) + "\n")
    r = subprocess.run([exe, path], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    return r.stdout.splitlines()


def test_ros_configuration_wiring_and_result(driver):
    import corenav_gp_amd.engine as engine
    g = load_golden("lookahead_restated")
    lines = run(driver, g, g["mean"], g["sigma"], float(g["arrival_time"]), float(g["now"]))
    # gp_predictor.cpp:11-13: same topics, service and queue sizes as the reference
    assert "subscribed /core_nav/core_nav/gp_result queue 1" in lines
    assert "advertised /core_nav/core_nav/stop_cmd queue 1" in lines
    assert "client /core_nav/core_nav/stopping_service" in lines
    assert "params_ok 1 service_calls 1 requested_stopping 1 clock_reads 2" in lines
    assert "published 1" in lines
    pub = [l for l in lines if l.startswith("publish ")][0].split()
    assert pub[1] == "/core_nav/core_nav/stop_cmd"
    assert float(pub[2]) == pytest.approx(float(g["stop_cmd"]), rel=1e-12)
    npub, cmd = engine.gppredictor_callback(g["mean"], g["sigma"], g["PvecData"], g["QvecData"], g["STMvecData"], g["HvecData"],
                                            g["PosData"], float(g["arrival_time"]), float(g["now"]))
    assert npub == 1 and float(pub[2]) == cmd          # both configurations of the class: identical, bitwise


def test_ros_configuration_publishes_nothing_on_an_empty_horizon(driver):
    g = load_golden("lookahead_restated")
    lines = run(driver, g, np.zeros(0), np.zeros(0), 0.0, 0.0)
    assert "published 0" in lines and "params_ok 1 service_calls 1 requested_stopping 1 clock_reads 2" in lines
