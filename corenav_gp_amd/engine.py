"""ctypes binding of include/corenav_gp.h (libcorenav_gp.so).  This is the reference-side stub
INTEGRATION.md shows for gp_slip_node.py; there is no CPU fallback -- a missing library or a
missing GPU raises."""
from __future__ import annotations

import ctypes
import os

import numpy as np

KERNEL_SE_ISO, KERNEL_SE_ARD, KERNEL_RBF_BROWNIAN = 0, 1, 2
F64, F32 = 0, 1
MAX_D = 8
MAX_THETA = MAX_D + 2
PROF_KERNELS = 5
PROF_NAMES = ("update", "potf2", "trmm", "finalize", "alpha")

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGP_LIB selects another build of the same ABI for measurements (the -DCGP_AB -DCGP_ABLATION library
# `make ab` writes next to the shipped one); there is still no CPU fallback.
LIB_PATH = os.environ.get("CGP_LIB") or os.path.join(_HERE, "libcorenav_gp.so")
DEBUG_SLOTS = 512
ABI_VERSION = 3   # include/corenav_gp.h CGP_ABI_VERSION: load() refuses a library of another revision
BUILD_ABLATION, BUILD_AB, BUILD_F32_NATIVE = 1, 2, 4
STREAM_CTX = ctypes.c_void_p(-1).value   # CGP_STREAM_CTX: the context's private stream

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
OBJECTIVE_FN = ctypes.CFUNCTYPE(ctypes.c_double, _dp, _dp, ctypes.c_int, ctypes.c_void_p)   # cgp_objective_fn

_SIGS = {
    "cgp_create": (_vp, [ctypes.c_int] * 6),
    "cgp_create_ex": (_vp, [ctypes.c_int] * 6 + [_ip]),
    "cgp_destroy": (None, [_vp]),
    "cgp_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "cgp_last_error": (ctypes.c_char_p, [_vp]),
    "cgp_abi_version": (ctypes.c_int, []),
    "cgp_build_flags": (ctypes.c_int, []),
    "cgp_synchronize": (ctypes.c_int, [_vp]),
    "cgp_sweep_create": (_vp, [_ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "cgp_sweep_destroy": (None, [_vp]),
    "cgp_sweep_ndev": (ctypes.c_int, [_vp]),
    "cgp_sweep_shard": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _ip, _ip]),
    "cgp_sweep_fit_predict": (ctypes.c_int, [_vp] + [ctypes.c_int] * 5 + [_dp, _dp, _dp, _dp, ctypes.c_int,
                                                                          ctypes.c_int, _dp, _dp, _dp, _ip, _dp]),
    "cgp_sweep_fit_predict_device": (ctypes.c_int, [_vp] + [ctypes.c_int] * 5 + [_vp] * 5 + [ctypes.c_int] + [_vp] * 5),
    "cgp_sweep_synchronize": (ctypes.c_int, [_vp]),
    "cgp_sweep_context": (_vp, [_vp, ctypes.c_int]),
    "cgp_fit": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp]),
    "cgp_predict": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp]),
    "cgp_get_alpha": (ctypes.c_int, [_vp, _dp]),
    "cgp_get_factor": (ctypes.c_int, [_vp, _dp]),
    "cgp_last_jitter": (ctypes.c_double, [_vp]),
    "cgp_nll_grad": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]),
    "cgp_optimize": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _ip]),
    "cgp_slip_node_callback_opt": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp,
                                                  ctypes.c_int, _ip]),
    "cgp_slip_node_callback": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                              ctypes.c_int, _ip]),
    "cgp_fit_predict_batch": (ctypes.c_int, [_vp] + [ctypes.c_int] * 5 + [_dp, _dp, _dp, _dp, ctypes.c_int,
                                                                          ctypes.c_int, _dp, _dp, _dp, _ip]),
    "cgp_fit_predict_batch_device": (ctypes.c_int, [_vp] + [ctypes.c_int] * 5 + [_vp, _vp, _vp, _vp, _vp,
                                                                                 ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cgp_window_init": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int]),
    "cgp_window_push": (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp, _dp]),
    "cgp_window_push_device": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "cgp_window_state": (ctypes.c_int, [_vp, ctypes.c_int, _ip, _ip]),
    "cgp_set_streams": (ctypes.c_int, [_vp, ctypes.c_int]),
    "cgp_set_refine": (ctypes.c_int, [_vp, ctypes.c_int]),
    "cgp_debug_read": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_longlong)]),
    "cgp_debug_buffers": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_ulonglong)]),
    "cgp_debug_small": (ctypes.c_int, [_vp, _dp]),
    "cgp_profile_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
    "cgp_profile_read": (ctypes.c_int, [_vp, _dp, _dp, ctypes.POINTER(ctypes.c_longlong)]),
    "cgp_llh_to_enu": (ctypes.c_int, [ctypes.c_double] * 3 + [_dp, _dp, _dp]),
    "cgp_predict_stop": (ctypes.c_int, [_dp, _dp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_int, _dp, _dp, _ip, _dp, _ip, _dp]),
    "cgp_predict_stop_batch": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int] + [_dp] * 9 + [ctypes.c_double, ctypes.c_int,
                                                                                            _dp, _dp, _ip, _dp, _ip, _dp]),
    "cgp_optimize_batch": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                          ctypes.c_int, ctypes.c_int, _dp, _ip]),
    "cgp_selftest_lbfgs": (ctypes.c_int, [_dp, ctypes.c_int, ctypes.c_int, _dp]),
    "cgp_lbfgs_minimize": (ctypes.c_int, [_vp, _vp, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                          _dp, _ip, _ip, _ip]),
    "cgp_recorder_create": (_vp, []),
    "cgp_recorder_destroy": (None, [_vp]),
    "cgp_recorder_update": (ctypes.c_int, [_vp, _dp, ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, ctypes.c_int, _ip]),
    "cgp_recorder_stop_cmd": (None, [_vp, ctypes.c_double]),
    "cgp_recorder_cmd": (None, [_vp, ctypes.c_double]),
    "cgp_recorder_state": (None, [_vp, _dp]),
    "cgp_gppredictor_callback": (ctypes.c_int, [_dp, _dp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_int, _ip, _dp]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Loads libcorenav_gp.so and binds every symbol of include/corenav_gp.h (raises if missing).
    Load order in a process that also uses torch.cuda: `import torch` FIRST -- torch bundles its own libamdhip64,
    and if this library has already pulled in /opt/rocm's copy, torch.cuda later reports "No HIP GPUs"."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.cgp_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} has ABI revision {lib.cgp_abi_version()}, this binding is for {ABI_VERSION}: rebuild it")
        _lib = lib
    return _lib


def lbfgs_minimize(fg, x0, max_evals=1000, pgtol=1e-5, factr=1e7):
    """cgp_lbfgs_minimize (host only): the engine's optimiser state machine on a Python objective fg(x) -> (f, grad).
    Returns (x, f, n_evals, n_iters, status)."""
    x = _d(x0).copy()
    n = len(x)

    def cb(xp, gp, nn, _user):
        f, g = fg(np.array([xp[i] for i in range(nn)]))
        for i in range(nn):
            gp[i] = g[i]
        return float(f)

    fn = OBJECTIVE_FN(cb)
    f = ctypes.c_double()
    nev, nit, st = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = load().cgp_lbfgs_minimize(ctypes.cast(fn, _vp), None, _p(x), n, max_evals, pgtol, factr, ctypes.byref(f),
                                   ctypes.byref(nev), ctypes.byref(nit), ctypes.byref(st))
    if rc != 0:
        raise CgpError(rc)
    return x, f.value, nev.value, nit.value, st.value


class CgpError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = load().cgp_strerror(code).decode()
        super().__init__(f"cgp error {code}: {msg}" + (f" [{detail}]" if detail else ""))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class Context:
    """One engine context (= one GPU, one stream).  Not thread-safe."""

    def __init__(self, device=0, max_n=2048, max_m=640, max_d=MAX_D, max_batch=1, dtype=F64):
        self.lib = load()
        self.dtype = dtype
        st = ctypes.c_int(0)
        self.h = self.lib.cgp_create_ex(device, max_n, max_m, max_d, max_batch, dtype, ctypes.byref(st))
        if not self.h:
            raise RuntimeError(f"cgp_create failed ({st.value}: {self.lib.cgp_strerror(st.value).decode()}): no usable gfx950 "
                               "device, an argument out of range, or out of device memory (the engine has no CPU fallback)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.cgp_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc < 0:
            raise CgpError(rc, self.lib.cgp_last_error(self.h).decode())
        return rc

    # -- single window -------------------------------------------------------------------------
    def fit(self, X, y, kernel_id, theta):
        X = _d(X)
        if X.ndim == 1:
            X = X[:, None]
        y, theta = _d(y).reshape(-1), _d(theta)
        logml = ctypes.c_double(0.0)
        rc = self._chk(self.lib.cgp_fit(self.h, _p(X), _p(y), X.shape[0], X.shape[1], kernel_id, _p(theta),
                                        ctypes.byref(logml)))
        self._n = X.shape[0]
        return rc, logml.value

    def predict(self, Xs, include_noise=True):
        Xs = _d(Xs)
        if Xs.ndim == 1:
            Xs = Xs[:, None]
        M = Xs.shape[0]
        mean, var = np.empty(M), np.empty(M)
        self._chk(self.lib.cgp_predict(self.h, _p(Xs), M, int(include_noise), _p(mean), _p(var)))
        return mean, var

    def alpha(self):
        a = np.empty(self._n)
        self._chk(self.lib.cgp_get_alpha(self.h, _p(a)))
        return a

    def factor(self):
        L = np.empty((self._n, self._n))
        self._chk(self.lib.cgp_get_factor(self.h, _p(L)))
        return L

    def last_jitter(self):
        return self.lib.cgp_last_jitter(self.h)

    def nll_grad(self, X, y, kernel_id, theta):
        """Negative log marginal likelihood and its gradient wrt the natural parameters."""
        X = _d(X)
        if X.ndim == 1:
            X = X[:, None]
        y, theta = _d(y).reshape(-1), _d(theta)
        nll, grad = ctypes.c_double(0.0), np.empty(len(theta))
        rc = self._chk(self.lib.cgp_nll_grad(self.h, _p(X), _p(y), X.shape[0], X.shape[1], kernel_id, _p(theta),
                                             ctypes.byref(nll), _p(grad)))
        if rc > 0:
            raise CgpError(rc)
        self._n = X.shape[0]
        return nll.value, grad

    def optimize(self, X, y, kernel_id, theta0, max_evals=1000):
        """GPy m.optimize() equivalent; returns (theta_opt, logml, n_evals)."""
        X = _d(X)
        if X.ndim == 1:
            X = X[:, None]
        y = _d(y).reshape(-1)
        theta = np.array(theta0, dtype=np.float64)
        logml, nev = ctypes.c_double(0.0), ctypes.c_int(0)
        rc = self._chk(self.lib.cgp_optimize(self.h, _p(X), _p(y), X.shape[0], X.shape[1], kernel_id, _p(theta),
                                             max_evals, ctypes.byref(logml), ctypes.byref(nev)))
        if rc > 0:
            raise CgpError(rc)
        self._n = X.shape[0]
        return theta, logml.value, nev.value

    def optimize_batch(self, X, y, kernel_id, theta0, max_evals=1000):
        """Batched m.optimize(): X (B, N, d), y (B, N), theta0 (B, nth) -> (theta_opt, logml, n_evals)."""
        X, y = _d(X), _d(y)
        B, N, d = X.shape
        theta = np.array(theta0, dtype=np.float64)
        if theta.ndim == 1:
            theta = np.tile(theta, (B, 1))
        theta = np.ascontiguousarray(theta)
        logml, nev = np.empty(B), np.zeros(B, dtype=np.int32)
        rc = self._chk(self.lib.cgp_optimize_batch(self.h, B, N, d, kernel_id, _p(X), _p(y), _p(theta), theta.shape[1],
                                                   max_evals, _p(logml), nev.ctypes.data_as(_ip)))
        if rc > 0:
            raise CgpError(rc)
        return theta, logml, nev

    def slip_node_callback_opt(self, time_array, slip_array, theta0, kernel_id=KERNEL_RBF_BROWNIAN, max_evals=1000,
                               cap=4096):
        t, s = _d(time_array).reshape(-1), _d(slip_array).reshape(-1)
        theta = np.array(theta0, dtype=np.float64)
        mean, sigma = np.empty(cap), np.empty(cap)
        m_out = ctypes.c_int(0)
        rc = self._chk(self.lib.cgp_slip_node_callback_opt(self.h, _p(t), _p(s), len(t), kernel_id, _p(theta), max_evals,
                                                           _p(mean), _p(sigma), cap, ctypes.byref(m_out)))
        if rc > 0:
            raise CgpError(rc)
        m = min(m_out.value, cap)
        return mean[:m].copy(), sigma[:m].copy(), theta

    def slip_node_callback(self, time_array, slip_array, theta, kernel_id=KERNEL_RBF_BROWNIAN, cap=4096):
        t, s, theta = _d(time_array).reshape(-1), _d(slip_array).reshape(-1), _d(theta)
        mean, sigma = np.empty(cap), np.empty(cap)
        m_out = ctypes.c_int(0)
        rc = self._chk(self.lib.cgp_slip_node_callback(self.h, _p(t), _p(s), len(t), kernel_id, _p(theta), _p(mean),
                                                       _p(sigma), cap, ctypes.byref(m_out)))
        if rc > 0:
            raise CgpError(rc)
        m = min(m_out.value, cap)
        return mean[:m].copy(), sigma[:m].copy()

    # -- batch, host buffers -------------------------------------------------------------------
    def fit_predict_batch(self, X, y, Xs, theta, kernel_id, include_noise=True):
        X, y, Xs, theta = _d(X), _d(y), _d(Xs), _d(theta)
        B, N, d = X.shape
        M = Xs.shape[1]
        mean, var = np.empty((B, M)), np.empty((B, M))
        logml, info = np.empty(B), np.zeros(B, dtype=np.int32)
        rc = self._chk(self.lib.cgp_fit_predict_batch(self.h, B, N, d, M, kernel_id, _p(X), _p(y), _p(Xs), _p(theta),
                                                      theta.shape[1], int(include_noise), _p(mean), _p(var),
                                                      _p(logml), info.ctypes.data_as(_ip)))
        return rc, mean, var, logml, info

    # -- batch, device pointers (ints) on a caller stream ------------------------------------------
    def fit_predict_batch_device(self, B, N, d, M, kernel_id, dX, dy, dXs, dtheta, djitter, include_noise, dmean,
                                 dvar, dlogml, dinfo, stream=0):
        """stream: a hipStream_t handle as an int.  0 is the legacy default stream itself (what
        torch.cuda.current_stream().cuda_stream returns for the default stream), so the work is ordered
        with the caller's default-stream kernels; engine.STREAM_CTX = the context's private stream."""
        return self._chk(self.lib.cgp_fit_predict_batch_device(self.h, B, N, d, M, kernel_id, dX, dy, dXs, dtheta,
                                                               djitter or None, int(include_noise), dmean, dvar,
                                                               dlogml, dinfo, ctypes.c_void_p(stream)))

    def synchronize(self):
        self._chk(self.lib.cgp_synchronize(self.h))

    # -- sliding windows (BASELINE configs[3]) ----------------------------------------------------
    def window_init(self, nwin, N, d, kernel_id, theta):
        theta = _d(theta)
        if theta.ndim == 1:
            theta = np.tile(theta, (nwin, 1))
        self._win = (nwin, d)
        self._chk(self.lib.cgp_window_init(self.h, nwin, N, d, kernel_id, _p(theta), theta.shape[1]))

    def window_push(self, xs, ys, include_noise=True):
        """xs (nwin, T, d), ys (nwin, T) -> one-step-ahead mean, variance and logML per tick, each (nwin, T)."""
        nwin, d = self._win
        xs, ys = _d(xs).reshape(nwin, -1, d), _d(ys).reshape(nwin, -1)
        T = ys.shape[1]
        pm, pv, lm = np.empty((nwin, T)), np.empty((nwin, T)), np.empty((nwin, T))
        rc = self._chk(self.lib.cgp_window_push(self.h, T, _p(xs), _p(ys), int(include_noise), _p(pm), _p(pv), _p(lm)))
        if rc > 0:
            raise CgpError(rc)
        return pm, pv, lm

    def window_push_device(self, T, dxs, dys, include_noise, dpm, dpv, dlm, stream=0):
        return self._chk(self.lib.cgp_window_push_device(self.h, T, dxs, dys, int(include_noise), dpm, dpv, dlm,
                                                         ctypes.c_void_p(stream)))

    def window_state(self, w=0):
        n, info = ctypes.c_int(0), ctypes.c_int(0)
        self._chk(self.lib.cgp_window_state(self.h, w, ctypes.byref(n), ctypes.byref(info)))
        return n.value, info.value

    def predict_stop_batch(self, mean, sigma, P, Q, STM, Hvec, pos, arrival, now, threshold=3.0, h_bug_compatible=True,
                           init_llh=None, init_ecef=None):
        """Batched GPU look-ahead (one wave per trajectory); returns (fired, stop_cmd, i, xy_err) arrays."""
        mean, sigma = _d(mean), _d(sigma)
        T, M = mean.shape
        P, Q, STM, Hvec, pos = (_d(a).reshape(T, -1) for a in (P, Q, STM, Hvec, pos))
        arrival, now = _d(np.broadcast_to(arrival, (T,))), _d(np.broadcast_to(now, (T,)))
        fired, iout = np.zeros(T, dtype=np.int32), np.zeros(T, dtype=np.int32)
        cmd, xy = np.zeros(T), np.zeros(T)
        illh, iecef = _d(init_llh if init_llh is not None else INIT_LLH), _d(init_ecef if init_ecef is not None else INIT_ECEF)
        self._chk(self.lib.cgp_predict_stop_batch(self.h, T, M, _p(mean), _p(sigma), _p(P), _p(Q), _p(STM), _p(Hvec),
                                                  _p(pos), _p(arrival), _p(now), threshold, int(h_bug_compatible),
                                                  _p(illh), _p(iecef), fired.ctypes.data_as(_ip), _p(cmd),
                                                  iout.ctypes.data_as(_ip), _p(xy)))
        return fired.astype(bool), cmd, iout, xy

    def debug_read(self):
        out = np.zeros(DEBUG_SLOTS, dtype=np.int64)
        self._chk(self.lib.cgp_debug_read(self.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))))
        return out

    def debug_buffers(self):
        """[(address, bytes)] of the context's large device buffers (include/corenav_gp.h: cgp_debug_buffers)."""
        out = np.zeros(16, dtype=np.uint64)
        self._chk(self.lib.cgp_debug_buffers(self.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))))
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(8)]

    def debug_small(self):
        """Raw result record of the last short-window launch (include/corenav_gp.h: cgp_debug_small)."""
        out = np.zeros(48)
        self._chk(self.lib.cgp_debug_small(self.h, _p(out)))
        return out

    def set_streams(self, n):
        self._chk(self.lib.cgp_set_streams(self.h, int(n)))

    def set_refine(self, steps=-1):
        """fp32 contexts: correction steps of alpha / the mean against a double-precision residual (cgp_set_refine): -1 the
        engine decides (one step: every fit at d <= 3, the dense fits beyond), 0 never, 1..3 always."""
        self._chk(self.lib.cgp_set_refine(self.h, int(steps)))

    def profile_enable(self, on=True):
        self._chk(self.lib.cgp_profile_enable(self.h, int(on)))

    def profile_read(self):
        ms, fl = np.zeros(PROF_KERNELS), np.zeros(PROF_KERNELS)
        n = np.zeros(PROF_KERNELS, dtype=np.int64)
        self._chk(self.lib.cgp_profile_read(self.h, _p(ms), _p(fl), n.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))))
        return {PROF_NAMES[i]: {"ms": float(ms[i]), "flops": float(fl[i]), "launches": int(n[i])}
                for i in range(PROF_KERNELS)}


class Sweep:
    """Multi-device sweep of independent fits through the C ABI (cgp_sweep_*): one context and one host
    thread per listed device, contiguous block partition, summaries gathered on the host."""

    def __init__(self, devices, max_n, max_m, max_d, max_batch_total, dtype=F64):
        self.lib = load()
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        self.h = self.lib.cgp_sweep_create(dv.ctypes.data_as(_ip), len(dv), max_n, max_m, max_d, max_batch_total, dtype)
        if not self.h:
            raise RuntimeError("cgp_sweep_create failed: a listed device is not a usable gfx950 GPU or out of memory")
        self.ndev = self.lib.cgp_sweep_ndev(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cgp_sweep_destroy(self.h)
            self.h = None

    __del__ = close

    def shard(self, batch, i):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        rc = self.lib.cgp_sweep_shard(self.h, batch, i, ctypes.byref(a), ctypes.byref(b))
        if rc:
            raise CgpError(rc)
        return a.value, b.value

    def fit_predict(self, X, y, Xs, theta, kernel_id, include_noise=True):
        X, y, Xs, theta = _d(X), _d(y), _d(Xs), _d(theta)
        B, N, d = X.shape
        M = Xs.shape[1]
        mean, var = np.empty((B, M)), np.empty((B, M))
        logml, info, summ = np.empty(B), np.zeros(B, dtype=np.int32), np.empty((B, 3))
        rc = self.lib.cgp_sweep_fit_predict(self.h, B, N, d, M, kernel_id, _p(X), _p(y), _p(Xs), _p(theta), theta.shape[1],
                                            int(include_noise), _p(mean), _p(var), _p(logml), info.ctypes.data_as(_ip),
                                            _p(summ))
        if rc < 0:
            raise CgpError(rc)
        return rc, mean, var, logml, info, summ

    def fit_predict_device(self, B, N, d, M, kernel_id, dX, dy, dXs, dtheta, djitter, include_noise, dmean, dvar, dlogml,
                           dinfo, streams=None):
        """cgp_sweep_fit_predict_device: every argument a list of ndev per-shard device pointers (ints), shard i's fits
        only; `streams` a list of hipStream_t handles (ints; None = the contexts' own streams).  Enqueues and returns."""
        def arr(ptrs):
            if ptrs is None:
                return None
            a = (ctypes.c_void_p * self.ndev)(*[ctypes.c_void_p(int(p) if p else 0) for p in ptrs])
            return ctypes.cast(a, _vp)
        keep = [arr(x) for x in (dX, dy, dXs, dtheta, djitter, dmean, dvar, dlogml, dinfo, streams)]
        rc = self.lib.cgp_sweep_fit_predict_device(self.h, B, N, d, M, kernel_id, keep[0], keep[1], keep[2], keep[3], keep[4],
                                                   int(include_noise), keep[5], keep[6], keep[7], keep[8], keep[9])
        if rc < 0:
            raise CgpError(rc)
        return rc

    def synchronize(self):
        rc = self.lib.cgp_sweep_synchronize(self.h)
        if rc:
            raise CgpError(rc)

    def set_streams(self, n):
        """cgp_set_streams on every shard's context."""
        for i in range(self.ndev):
            rc = self.lib.cgp_set_streams(self.lib.cgp_sweep_context(self.h, i), n)
            if rc:
                raise CgpError(rc)

    def set_refine(self, steps=-1):
        """cgp_set_refine on every shard's context."""
        for i in range(self.ndev):
            rc = self.lib.cgp_set_refine(self.lib.cgp_sweep_context(self.h, i), int(steps))
            if rc:
                raise CgpError(rc)


INIT_LLH = (0.693457963620326, -1.39498384275845, 334.993517334743)   # init_params.yaml:13-16
INIT_ECEF = (859153.0153, -4836303.7266, 4055378.501)                  # init_params.yaml:9-12


def llh_to_enu(lat, lon, h, init_llh=INIT_LLH, init_ecef=INIT_ECEF):
    out = np.empty(3)
    rc = load().cgp_llh_to_enu(lat, lon, h, _p(_d(init_llh)), _p(_d(init_ecef)), _p(out))
    if rc:
        raise CgpError(rc)
    return out


def predict_stop(mean, sigma, PvecData, QvecData, STMvecData, HvecData, pos_llh, arrival_time=0.0, now=0.0,
                 threshold=3.0, h_bug_compatible=True, init_llh=INIT_LLH, init_ecef=INIT_ECEF):
    mean, sigma = _d(mean), _d(sigma)
    fired, i = ctypes.c_int(0), ctypes.c_int(0)
    cmd, xy = ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = load().cgp_predict_stop(_p(mean), _p(sigma), len(mean), _p(_d(PvecData)), _p(_d(QvecData)),
                                 _p(_d(STMvecData)), _p(_d(HvecData)), _p(_d(pos_llh)), arrival_time, now, threshold,
                                 int(h_bug_compatible), _p(_d(init_llh)), _p(_d(init_ecef)), ctypes.byref(fired),
                                 ctypes.byref(cmd), ctypes.byref(i), ctypes.byref(xy))
    if rc:
        raise CgpError(rc)
    return bool(fired.value), cmd.value, i.value, xy.value


def gppredictor_callback(mean, sigma, PvecData, QvecData, STMvecData, HvecData, pos_llh, arrival_time, now,
                         h_bug_compatible=True):
    """One GpPredictor::GPCallBack through the C++ class; returns (n_published, stop_cmd)."""
    mean, sigma = _d(mean), _d(sigma)
    npub, cmd = ctypes.c_int(0), ctypes.c_double(0.0)
    rc = load().cgp_gppredictor_callback(_p(mean), _p(sigma), len(mean), _p(_d(PvecData)), _p(_d(QvecData)),
                                         _p(_d(STMvecData)), _p(_d(HvecData)), _p(_d(pos_llh)), arrival_time, now,
                                         int(h_bug_compatible), ctypes.byref(npub), ctypes.byref(cmd))
    if rc:
        raise CgpError(rc)
    return npub.value, cmd.value


class SlipRecorder:
    """CoreNav's slip computation + recording-window state machine (C++ class behind the C ABI)."""
    STATE_FIELDS = ("odomUptCount", "startRecording", "stopRecording", "gp_flag", "first_driving_flag",
                    "new_stop_data_arrived_", "skipped_windows", "cmd_stop_")

    def __init__(self):
        self.lib = load()
        self.h = self.lib.cgp_recorder_create()
        self.slip = 0.0

    def close(self):
        if getattr(self, "h", None):
            self.lib.cgp_recorder_destroy(self.h)
            self.h = None

    __del__ = close

    def update(self, vfl, vfr, vbl, vbr, vlin, cmd_x, cap=256):
        wv = _d([vfl, vfr, vbl, vbr])
        slip, n = ctypes.c_double(0.0), ctypes.c_int(0)
        t, s = np.empty(cap), np.empty(cap)
        pub = self.lib.cgp_recorder_update(self.h, _p(wv), vlin, cmd_x, ctypes.byref(slip), _p(t), _p(s), cap,
                                           ctypes.byref(n))
        self.slip = slip.value
        if pub < 0:
            raise CgpError(pub)
        return (t[:n.value].copy(), s[:n.value].copy()) if pub == 1 else None

    def stop_callback(self, cmd_stop):
        self.lib.cgp_recorder_stop_cmd(self.h, float(cmd_stop))

    def cmd_callback(self, cmd_x):
        self.lib.cgp_recorder_cmd(self.h, float(cmd_x))

    def state(self):
        st = np.zeros(8)
        self.lib.cgp_recorder_state(self.h, _p(st))
        return dict(zip(self.STATE_FIELDS, st))
