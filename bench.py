#!/usr/bin/env python3
"""bench.py -- GP-fits/s of the slip-GP hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic slip windows: `--batch`
independent fixed-theta fits (Gram + Cholesky + solve + predictive mean/variance + log marginal
likelihood, M = 599 test points) per GPU, inputs already resident in HBM.  At N = 1 the workload is
BASELINE configs[1] (N = 2048, d = 6, ARD, fp64).  Multi-GPU: one process per GPU (torchrun), fits
sharded across ranks with no data-path collective; RCCL only gathers per-fit summaries.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (= vector) peak; absent from the local
                               # guide, re-measured by tools/mfma_peak (see DESIGN.md "Measurement")
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
M_TEST = 599                   # gp_slip_node.py:45,59: arange(min, max+600)[n:] -> 599 points


def fit_flops(N, d, M):
    """Algorithmic flops of one fit: F_chol and F_fit."""
    f_chol = N ** 3 / 3.0
    f_fit = f_chol + M * N ** 2 + 2 * N ** 2 + (3 * d + 2) * (N * (N + 1) / 2 + M * N) + 4 * M * N
    return f_chol, f_fit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="independent fits per GPU per step (2 per CU)")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3])
    ap.add_argument("--n", type=int, default=None, help="override window length N")
    ap.add_argument("--cpu-sample", type=int, default=12, help="fits timed on the host for cpu_baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="worker streams the batch is spread over")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import corenav_gp_amd.engine as engine
    import corenav_gp_amd.synth as synth
    from corenav_gp_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # CGP_BENCH_BACKEND=gloo + CGP_BENCH_SAME_DEVICE=1: self-test of the N > 1 flow on a 1-GPU box
    backend = os.environ.get("CGP_BENCH_BACKEND", "nccl")
    if os.environ.get("CGP_BENCH_SAME_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # where collective tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.batch
    # every rank owns its own shard of windows (weak scaling: per-GPU work fixed)
    kid, X, y, Xs, th, dts = synth.config(args.config, batch=B, N=args.n, M=M_TEST)
    if rank > 0:   # different trajectories per rank, same shapes
        X = np.roll(X, rank, axis=0) + 0.0
        rng = np.random.default_rng(synth.SEED_BASE + 7919 * rank)
        y = y + rng.normal(0, 1e-3, y.shape)
    _, N, d = X.shape
    dtype = engine.F64 if dts == "f64" else engine.F32
    tdt = torch.float64 if dts == "f64" else torch.float32
    nth = th.shape[1]
    thp = np.zeros((B, engine.MAX_THETA))
    thp[:, :nth] = th

    dX = torch.from_numpy(np.ascontiguousarray(X.transpose(0, 2, 1))).to(dev, tdt)     # [B][d][N]
    dXs = torch.from_numpy(np.ascontiguousarray(Xs.transpose(0, 2, 1))).to(dev, tdt)   # [B][d][M]
    dy = torch.from_numpy(y).to(dev, tdt)
    dth = torch.from_numpy(thp).to(dev, torch.float64)
    dmean = torch.empty((B, M_TEST), device=dev, dtype=tdt)
    dvar = torch.empty((B, M_TEST), device=dev, dtype=tdt)
    dlogml = torch.empty(B, device=dev, dtype=torch.float64)
    dinfo = torch.zeros(B, device=dev, dtype=torch.int32)

    ctx = engine.Context(device=local, max_n=N, max_m=M_TEST, max_d=d, max_batch=B, dtype=dtype)
    ctx.set_streams(args.streams)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.fit_predict_batch_device(B, N, d, M_TEST, kid, dX.data_ptr(), dy.data_ptr(), dXs.data_ptr(),
                                     dth.data_ptr(), 0, True, dmean.data_ptr(), dvar.data_ptr(), dlogml.data_ptr(),
                                     dinfo.data_ptr(), stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=cdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if not os.environ.get("CGP_DBG"):   # timing ablations produce wrong factors on purpose
        assert int(dinfo.abs().sum().item()) == 0, "a fit reported a non-positive pivot"

    # latency of ONE fit of the same shape (BASELINE configs[1] reads "single GP fit"): the engine
    # switches to its split-K latency schedule for <= 4 fits; synchronised per call
    def one():
        ctx.fit_predict_batch_device(1, N, d, M_TEST, kid, dX.data_ptr(), dy.data_ptr(), dXs.data_ptr(),
                                     dth.data_ptr(), 0, True, dmean.data_ptr(), dvar.data_ptr(), dlogml.data_ptr(),
                                     dinfo.data_ptr(), stream)
        torch.cuda.synchronize()
    keep = (dmean[0].clone(), dvar[0].clone(), dlogml[0].clone())
    for _ in range(3):
        one()
    t1 = time.perf_counter()
    for _ in range(10):
        one()
    single_ms = (time.perf_counter() - t1) / 10 * 1e3
    dmean[0], dvar[0], dlogml[0] = keep      # the batch outputs are what the summaries / oracle check read

    # per-fit summaries gathered over RCCL (the only collective on the path: SURVEY.md 8e)
    summ = torch.stack([dlogml, 2.0 * dvar.to(torch.float64).max(1).values.sqrt(), dinfo.to(torch.float64)], 1)
    table = sharding.gather_summaries(summ.to(cdev), B * world)
    ens = sharding.ensemble_stats(table)

    # ---- roofline of the dominant kernel (k_panel: trailing syrk/gemm + Gram + trmm), HIP events per launch
    ctx.profile_enable(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    upd = prof["update"]
    achieved = upd["flops"] / (upd["ms"] * 1e-3) / 1e12 if upd["ms"] > 0 else 0.0
    peak = FP64_MFMA_PEAK_TFLOPS if dts == "f64" else FP32_MFMA_PEAK_TFLOPS

    if rank == 0:
        fits = B * world * args.steps
        f_chol, f_fit = fit_flops(N, d, M_TEST)
        value = fits / dt
        out = {
            "metric": "GP-fits/s", "value": value, "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dts, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: batch of independent fixed-theta GP fits, "
                                   f"N={N} d={d} M={M_TEST} kernel={'SE-ARD' if kid == 1 else 'SE-iso'} {dts}",
                       "fits_per_gpu_per_step": B, "streams": args.streams, "N": N, "d": d, "M": M_TEST,
                       "single_fit_latency_ms": single_ms,
                       "fit_tflops": value * f_fit / 1e12, "cholesky_roofline_frac": value * f_chol / 1e12 / world / peak,
                       "inputs": "resident in HBM", "ensemble": ens},
            "roofline": {"bound": "mfma", "kernel": "k_panel (syrk/gemm trailing update + fused Gram + in-register trmm)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic(B, N, dts), "avg_launch_ms": upd["ms"] / max(upd["launches"], 1),
                         "launches": upd["launches"]},
            "kernel_ms_per_step": {k: v["ms"] / 2 for k, v in prof.items()},
        }
        if not args.no_cpu and world == 1:   # rank 0 at N = 1 only: other ranks would wait on the host work
            out["cpu_baseline"] = cpu_baseline(kid, X, y, Xs, th, args.cpu_sample, dmean, dvar, dlogml, f_fit)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(B, N, dts):
    """HBM bytes per k_update launch from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs, FETCH doubled for the gfx950 half-count).  Only valid
    for the configuration the profile was taken on (default workload); otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        w = d.get("_workload", {"batch": 256, "N": 2048, "dtype": "f64"})
        if (B, N, dts) != (w["batch"], w["N"], w["dtype"]):
            return None
        return d.get("k_panel", d.get("k_update"))["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(kid, X, y, Xs, th, nsample, dmean, dvar, dlogml, f_fit):
    """The C oracle ('port' of the reference arithmetic, single thread like the reference's catkin
    build) timed on this box's host cores on a bounded sample of the same windows; its outputs also
    check the timed GPU outputs."""
    import ctypes
    import subprocess
    # compiled HERE (-march=native of the box that runs the baseline), not shipped from the build container
    import tempfile
    so = os.path.join(tempfile.mkdtemp(prefix="cgp_oracle_"), "libgp_oracle.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "gp_oracle.c"), "-lm"])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    n = min(nsample, X.shape[0])
    N, d = X.shape[1:]
    M = Xs.shape[1]
    gm, gv, gl = dmean.cpu().numpy().astype(np.float64), dvar.cpu().numpy().astype(np.float64), dlogml.cpu().numpy()
    worst = 0.0
    t0 = time.perf_counter()
    for b in range(n):
        mean, var, logml, jit = np.zeros(M), np.zeros(M), np.zeros(1), np.zeros(1)
        p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
        Xb, yb, Xsb, thb = (np.ascontiguousarray(a[b], dtype=np.float64) for a in (X, y, Xs, th))
        rc = lib.oracle_fit_predict(kid, p(thb), N, d, p(Xb), p(yb), M, p(Xsb), 1, p(mean), p(var), p(logml), None,
                                    None, p(jit))
        assert rc == 0
        worst = max(worst, float(np.max(np.abs(gm[b] - mean)) / np.max(np.abs(mean))),
                    float(np.max(np.abs(gv[b] - var) / var)), abs(gl[b] - logml[0]) / abs(logml[0]))
    el = time.perf_counter() - t0
    tol = 1e-6 if str(dmean.dtype).endswith("float64") else 1e-3   # north_star parity bar
    assert worst < tol, f"timed GPU outputs differ from the oracle: max rel err {worst:.3e} >= {tol}"
    # extra context row (not the baseline): the same C port on every host core at once, one window per
    # thread (ctypes releases the GIL), i.e. the batched workload as a many-core host would run it
    allc = None
    try:
        from concurrent.futures import ThreadPoolExecutor
        try:
            ncore = len(os.sched_getaffinity(0))
        except Exception:
            ncore = os.cpu_count() or 1
        ncore = max(1, min(ncore, 64))   # bounded: the visible core count can exceed the container's CPU quota

        def one(i):
            b = i % X.shape[0]
            mean, var, logml, jit = np.zeros(M), np.zeros(M), np.zeros(1), np.zeros(1)
            q = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
            Xb, yb, Xsb, thb = (np.ascontiguousarray(a[b], dtype=np.float64) for a in (X, y, Xs, th))
            return lib.oracle_fit_predict(kid, q(thb), N, d, q(Xb), q(yb), M, q(Xsb), 1, q(mean), q(var), q(logml),
                                          None, None, q(jit))
        with ThreadPoolExecutor(max_workers=ncore) as ex:
            t2 = time.perf_counter()
            rcs = list(ex.map(one, range(ncore)))
            el2 = time.perf_counter() - t2
        assert not any(rcs)
        allc = {"value": ncore / el2, "unit": "fits/s", "cores": ncore, "kind": "port, one window per thread (threads, not necessarily physical cores: container CPU quota applies)",
                "sample": f"{ncore} windows, {el2:.1f} s"}
    except Exception as e:
        allc = {"error": repr(e)}
    # extra context row (not the baseline): the numpy/scipy restatement, LAPACK on all host cores
    lap = None
    try:
        from oracle import gp_oracle as go
        t1 = time.perf_counter()
        nl = min(2, n)
        for b in range(nl):
            f = go.fit(kid, th[b], X[b], y[b])
            go.predict(f, Xs[b])
        lap = {"value": nl / (time.perf_counter() - t1), "unit": "fits/s", "cores": os.cpu_count(),
               "kind": "numpy/scipy LAPACK restatement, all cores", "sample": f"{nl} windows"}
    except Exception as e:   # the baseline proper does not depend on it
        lap = {"error": repr(e)}
    return {"value": n / el, "unit": "fits/s", "cores": 1, "kind": "port", "port_all_cores": allc, "lapack_all_cores": lap,
            "sample": f"{n} of the step's windows through oracle/gp_oracle.c (gcc -O3 -march=native, 1 thread), "
                      f"{el:.1f} s; {n / el * f_fit / 1e9:.2f} GFLOP/s",
            "gpu_vs_oracle_max_rel_err": worst, "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
