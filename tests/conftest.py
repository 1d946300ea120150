import os
import subprocess
import sys

import numpy as np
import pytest
try:  # FIRST when it exists: torch bundles its own libamdhip64; if libcorenav_gp.so pulled in /opt/rocm's copy before
    #   torch is loaded, torch.cuda would later find "No HIP GPUs" in the same process (see engine.load()).  The pure
    #   host / oracle / ABI tests do not need torch.
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle on oracle/libgp_oracle.so (built on demand with gcc; checker only)."""
    import ctypes
    so = os.path.join(ROOT, "oracle", "libgp_oracle.so")
    src = os.path.join(ROOT, "oracle", "gp_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_fit_predict.restype = ctypes.c_int
    lib.oracle_fit_predict.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int, dp,
                                       ctypes.c_int, dp, dp, dp, dp, dp, dp]

    def fit_predict(kid, theta, X, y, Xs, include_noise=True):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        N, d = X.shape
        M = Xs.shape[0]
        mean, var, logml = np.zeros(M), np.zeros(M), np.zeros(1)
        alpha, jit = np.zeros(N), np.zeros(1)
        p = lambda a: a.ctypes.data_as(dp)
        rc = lib.oracle_fit_predict(kid, p(theta), N, d, p(X), p(y), M, p(Xs), int(include_noise), p(mean), p(var),
                                    p(logml), p(alpha), None, p(jit))
        return rc, mean, var, float(logml[0]), alpha, float(jit[0])

    return fit_predict


def openblas_path():
    """The OpenBLAS that scipy bundles (single-thread LAPACK row of the CPU baseline); None if there is none."""
    import glob
    try:
        import scipy
    except ImportError:
        return None
    hits = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so")))
    return hits[0] if hits else None


@pytest.fixture(scope="session")
def oracle_c_lapack():
    """ctypes handle on oracle/libgp_oracle_lapack.so (checker / CPU baseline only)."""
    import ctypes
    path = openblas_path()
    if path is None:
        pytest.skip("no bundled OpenBLAS found")
    so = os.path.join(ROOT, "oracle", "libgp_oracle_lapack.so")
    src = os.path.join(ROOT, "oracle", "gp_oracle_lapack.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_lapack_init.restype = ctypes.c_int
    lib.oracle_lapack_init.argtypes = [ctypes.c_char_p]
    assert lib.oracle_lapack_init(path.encode()) == 0
    lib.oracle_fit_predict_lapack.restype = ctypes.c_int
    lib.oracle_fit_predict_lapack.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int, dp,
                                              ctypes.c_int, dp, dp, dp, dp, dp, dp]

    def fit_predict(kid, theta, X, y, Xs, include_noise=True):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        N, d = X.shape
        M = Xs.shape[0]
        mean, var, logml = np.zeros(M), np.zeros(M), np.zeros(1)
        alpha, jit = np.zeros(N), np.zeros(1)
        p = lambda a: a.ctypes.data_as(dp)
        rc = lib.oracle_fit_predict_lapack(kid, p(theta), N, d, p(X), p(y), M, p(Xs), int(include_noise), p(mean), p(var),
                                           p(logml), p(alpha), p(jit), None)
        return rc, mean, var, float(logml[0]), alpha, float(jit[0])

    return fit_predict
