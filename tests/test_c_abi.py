"""A compiled C / C++ caller of the C ABI (north_star: "host code stays C++ ... through a thin C-ABI layer"; the reference's
real caller is a compiled node, gp_predictor/src/gp_predictor.cpp:180-190).  CPU: include/corenav_gp.h is valid ISO C11
(`gcc -std=c11 -pedantic -Werror`) and valid C++17, and both callers link against libcorenav_gp.so.  GPU: the binaries run
a fit, a prediction, a batch, a two-shard sweep and the node callback on golden fixtures dumped to .bin files, and check
the values themselves (no Python in the loop but the fixture dump)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

HERE = os.path.join(ROOT, "tests", "c_abi")
LIBDIR = os.path.join(ROOT, "corenav_gp_amd")


def _build(tmp, src, cc, std):
    exe = os.path.join(tmp, os.path.basename(src).replace(".", "_"))
    cmd = [cc, std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
           "-L", LIBDIR, "-lcorenav_gp", "-lm", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


@pytest.fixture(scope="module")
def binaries(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "libcorenav_gp.so")):
        import __graft_entry__ as ge
        ge.build()
    tmp = str(tmp_path_factory.mktemp("c_abi"))
    return tmp, _build(tmp, os.path.join(HERE, "caller.c"), "gcc", "-std=c11"), _build(tmp, os.path.join(HERE, "caller.cpp"), "g++", "-std=c++17")


def test_header_is_iso_c_and_callers_link(binaries):
    tmp, c_exe, cpp_exe = binaries
    assert os.access(c_exe, os.X_OK) and os.access(cpp_exe, os.X_OK)
    # the header alone, as C89-compatible-comment C11 and as C++: no declaration needs anything but <stddef.h>
    for cc, std, lang in (("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "c++")):
        subprocess.check_call([cc, std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", lang,
                               os.path.join(ROOT, "include", "corenav_gp.h")])


def _dump_fit_fixture(path, name):
    g = load_golden(name)
    X, Xs, th = np.asarray(g["X"], float), np.asarray(g["Xs"], float), np.asarray(g["theta"], float)
    N, d = X.shape
    hdr = np.array([N, d, Xs.shape[0], int(g["kernel_id"]), len(th)], dtype=np.float64)
    parts = [hdr, th, X.ravel(), g["y"], Xs.ravel(), g["mean"], g["var_latent"], np.array([float(g["logml"])]), g["alpha"]]
    np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in parts]).tofile(path)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("sk_se_ard_n256_d6", "fp64"), ("sk_se_iso_n256_d3", "fp64"), ("sk_se_ard_n256_d6", "fp32")])
def test_c_caller_runs_a_fit_on_the_gpu(binaries, name, mode):
    tmp, c_exe, _ = binaries
    fx = os.path.join(tmp, name + ".bin")
    _dump_fit_fixture(fx, name)
    r = subprocess.run([c_exe, fx] + (["fp32"] if mode == "fp32" else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "caller.c ok" in r.stdout


@pytest.mark.gpu
def test_cpp_caller_runs_the_node_callback_on_the_gpu(binaries):
    tmp, _, cpp_exe = binaries
    g = load_golden("slipval_window_rbfbrownian")
    t, s, th = np.asarray(g["time_array"], float), np.asarray(g["slip_array"], float), np.asarray(g["theta"], float)
    em, es = np.asarray(g["mean"], float), np.asarray(g["sigma"], float)   # the fixture's published arrays (gp_slip_node.py:59-61)
    fx = os.path.join(tmp, "window.bin")
    np.concatenate([[len(t)], t, s, th, [len(em)], em, es]).astype(np.float64).tofile(fx)
    r = subprocess.run([cpp_exe, fx], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "caller.cpp ok" in r.stdout
