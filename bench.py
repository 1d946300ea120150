#!/usr/bin/env python3
"""bench.py -- GP-fits/s of the slip-GP hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic slip windows: `--batch`
independent fixed-theta fits (Gram + Cholesky + solve + predictive mean/variance + log marginal
likelihood, M = 599 test points) per GPU, inputs already resident in HBM.  At N = 1 the workload is
BASELINE configs[1] (N = 2048, d = 6, ARD, fp64).

Multi-GPU: one process per GPU, fits sharded across ranks with no data-path collective; RCCL only
gathers per-fit summaries.  `python bench.py --gpus N` on its own STARTS the N ranks (fresh child
processes, before this process has touched the GPU) and relays rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
Default: weak scaling (every rank owns `--batch` fits).  `--scaling strong` fixes the TOTAL: `--batch` fits
per step cut into contiguous per-rank shards (sharding.shard_range) -- `--scaling strong --config 3` is BASELINE
configs[2] as written ("Batch=512 independent N=1024 GP fits fp32 sharded across 8 MI355X": 64 fits per GPU
and call at --gpus 8).  CGP_BENCH_FORCE_DIST=1 runs the RCCL rendezvous / barrier / all_gather / all_reduce
with a world of ONE rank (the collective path on a one-GPU box).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (= vector) peak; absent from the local
                               # guide, re-measured by tools/mfma_peak (see DESIGN.md "Measurement")
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (same guide; the 2:1-sparsity headline figure is twice this)
HBM_PEAK_GBPS = 8000.0         # same guide, "HBM3E peak BW" (spec)
M_TEST = 599                   # gp_slip_node.py:45,59: arange(min, max+600)[n:] -> 599 points


def fit_flops(N, d, M):
    """Algorithmic flops of one fit: F_chol and F_fit."""
    f_chol = N ** 3 / 3.0
    f_fit = f_chol + M * N ** 2 + 2 * N ** 2 + (3 * d + 2) * (N * (N + 1) / 2 + M * N) + 4 * M * N
    return f_chol, f_fit


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="independent fits per GPU per step (2 per CU)")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3])
    ap.add_argument("--n", type=int, default=None, help="override window length N")
    ap.add_argument("--cpu-sample", type=int, default=0, help="fits timed on the host for cpu_baseline (0 = by time budget)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip config.extra (cfg3 / window / look-ahead / end-to-end lines)")
    ap.add_argument("--streams", type=int, default=0, help="cgp_set_streams: 0 = the engine decides (default), 1 = one group, n = n groups")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="contexts (slab sets) the steps are dealt over, one HIP stream each, so that successive calls too small "
                         "to fill the chip overlap on the GPU (cross-sweep overlap: a throughput-of-repeated-sweeps figure, not the "
                         "speed-up of one sweep); 0 = 1, the single-context figure")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch fits per GPU and step; strong: --batch fits per step over ALL ranks (contiguous shards)")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`bench.py --gpus N` outside a launcher: start N fresh rank processes (this process has made no
    HIP / torch.cuda call and never will), one per GPU, relay rank 0's JSON line and the worst exit
    code.  Never re-execs a process that has touched the GPU.  Rendezvous through a file store (no port to
    race for); every child is polled, and as soon as one exits non-zero -- or the global deadline passes -- the
    others are killed (exact PIDs) and that code is returned, so a rank that dies at start-up does not leave the
    rest waiting in init_process_group."""
    import tempfile
    rdzv = tempfile.NamedTemporaryFile(prefix="cgp_rdzv_", dir="/tmp", delete=False)
    rdzv.close()
    os.unlink(rdzv.name)        # FileStore creates it
    deadline = time.time() + float(os.environ.get("CGP_BENCH_SPAWN_TIMEOUT", "1500"))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   CGP_BENCH_INIT="file://" + rdzv.name)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the pool's driver only supports dmabuf IPC (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    out0 = []
    t0 = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # drain rank 0's pipe while polling
    t0.start()
    rcs = [None] * args.gpus
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = rcs[r]
        if failed is not None or time.time() > deadline:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()
                    rcs[r] = p.wait()
            if failed is None:
                failed = -9
            break
        time.sleep(0.05)
    t0.join(timeout=10)
    try:
        os.unlink(rdzv.name)
    except OSError:
        pass
    lines = [l for l in ("".join(out0)).splitlines() if l.startswith("{")]
    if failed is not None or not lines:
        sys.stderr.write(f"bench.py: rank exit codes {rcs}" + ("; no result line\n" if not lines else "\n"))
        return failed if failed else 1
    print(lines[-1])
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run_rank(args)


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import corenav_gp_amd.synth as synth
    from corenav_gp_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CGP_BENCH_TEST_DIE_RANK") == str(rank):   # launcher self-test: this rank fails before the rendezvous
        raise SystemExit(17)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: every rank must be started with the same --gpus")
    # CGP_BENCH_FORCE_DIST=1: the collective path (rendezvous, barrier, all_gather, all_reduce, the summary gather)
    # with a world of one rank -- RCCL loads and runs on a one-GPU box before an 8-GPU job is the first to try it
    use_dist = world > 1 or bool(os.environ.get("CGP_BENCH_FORCE_DIST"))
    # CGP_BENCH_BACKEND=gloo + CGP_BENCH_SAME_DEVICE=1: self-test of the N > 1 flow on a 1-GPU box.
    # CGP_BENCH_DRY=1 (CPU test of the launcher / rendezvous / gather only): no engine, no GPU, the
    # "step" is a sleep and the line says "dry_run": true -- never a measurement.
    backend = os.environ.get("CGP_BENCH_BACKEND", "nccl")
    dry = bool(os.environ.get("CGP_BENCH_DRY"))
    if os.environ.get("CGP_BENCH_SAME_DEVICE"):
        local = 0
    dev = torch.device("cpu")
    if not dry:
        import corenav_gp_amd.engine as engine
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # where collective tensors live
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {"device_id": dev} if backend == "nccl" else {}
        init = os.environ.get("CGP_BENCH_INIT")       # spawn_ranks: file store
        if init is None and "MASTER_PORT" not in os.environ:   # forced one-rank world outside any launcher
            import tempfile
            f = tempfile.NamedTemporaryFile(prefix="cgp_rdzv_", dir="/tmp", delete=False)
            f.close()
            os.unlink(f.name)
            init = "file://" + f.name
        if init is not None:
            dist.init_process_group(backend, init_method=init, rank=rank, world_size=world, **kw)
        else:                                         # torchrun / the driver's launcher: env:// rendezvous
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend, **kw)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    strong = args.scaling == "strong"
    nsyn = (lambda b: min(b, 4)) if dry else (lambda b: b)     # dry run: the windows are never fitted, a few suffice
    if strong:
        # the sweep's total is fixed: this rank owns the contiguous shard [b0, b1) of the args.batch windows
        B_total = args.batch
        b0, b1 = sharding.shard_range(B_total, rank, world)
        B = b1 - b0
        if B < 1:
            raise SystemExit(f"--scaling strong: {args.batch} fits cannot be cut into {world} non-empty shards")
        kid, X, y, Xs, th, dts = synth.config(args.config, batch=nsyn(B), N=args.n if not dry else 64, M=M_TEST, first=b0)
    else:
        B = args.batch
        B_total = B * world
        # every rank owns its own batch of windows (weak scaling: per-GPU work fixed)
        kid, X, y, Xs, th, dts = synth.config(args.config, batch=nsyn(B), N=args.n if not dry else 64, M=M_TEST)
        if rank > 0:   # different trajectories per rank, same shapes
            X = np.roll(X, rank, axis=0) + 0.0
            rng = np.random.default_rng(synth.SEED_BASE + 7919 * rank)
            y = y + rng.normal(0, 1e-3, y.shape)
    _, N, d = X.shape
    peak = FP64_MFMA_PEAK_TFLOPS if dts == "f64" else FP32_MFMA_PEAK_TFLOPS
    ablation = False
    depth = args.pipeline if args.pipeline > 0 else 1
    if not dry:
        W = Workload(engine, torch, dev, local, kid, X, y, Xs, th, dts, args.streams)
        ablation = bool(engine.load().cgp_build_flags() & engine.BUILD_ABLATION) and bool(os.environ.get("CGP_DBG"))
        step = W.step
        if depth > 1:
            # Successive calls of a sweep are independent; a call of a few dozen fits leaves most CUs idle in its
            # chain-bound early block steps.  `depth` contexts (own slabs, own outputs), one HIP stream each, take
            # the steps round-robin: step i + 1's early launches run beside step i's MFMA-bound late ones.  Every
            # step still does all of its work; the timed region is bracketed as before.
            lanes = [W] + [Workload(engine, torch, dev, local, kid, X, y, Xs, th, dts, args.streams) for _ in range(depth - 1)]
            for ln in lanes:
                ln.torch_stream = torch.cuda.Stream(dev)
                ln.stream = ln.torch_stream.cuda_stream
            torch.cuda.synchronize()
            turn = [0]

            def step():
                lanes[turn[0] % depth].step()
                turn[0] += 1
    else:
        def step():
            time.sleep(0.002)

    def fence():
        if use_dist:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    dt_local = time.perf_counter() - t0
    tall = torch.tensor([dt_local, float(B)], device=cdev, dtype=torch.float64)
    per_rank = [tall.clone() for _ in range(world)]
    if use_dist:
        dist.all_gather(per_rank, tall)
        tmax = tall[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tall[0] = tmax[0]
    dt = float(tall[0].item())
    per_rank_fits = [float(t[1].item()) * args.steps / float(t[0].item()) for t in per_rank]
    if not dry and not ablation:   # timing ablations (-DCGP_ABLATION build + CGP_DBG) produce wrong factors on purpose
        assert int(W.dinfo.abs().sum().item()) == 0, "a fit reported a non-positive pivot"

    single_ms = None
    if not dry:
        if depth > 1:
            assert all(int(ln.dinfo.abs().sum().item()) == 0 for ln in lanes)
            W.stream = torch.cuda.current_stream().cuda_stream   # what follows is lane 0 alone, on the default stream
            torch.cuda.synchronize()
        single_ms = W.single_fit_latency_ms()
        # per-fit summaries gathered over RCCL (the only collective on the path: SURVEY.md 8e)
        summ = torch.stack([W.dlogml, 2.0 * W.dvar.to(torch.float64).max(1).values.sqrt(), W.dinfo.to(torch.float64)], 1)
    else:
        summ = torch.zeros((B, 3), dtype=torch.float64)
    table = sharding.gather_summaries(summ.to(cdev), B_total)
    ens = sharding.ensemble_stats(table)
    assert table.shape[0] == B_total

    # A plain `bench.py --gpus N` (weak scaling of the headline) also carries BASELINE configs[2] AS WRITTEN -- 512 x N=1024
    # fp32 fits cut into contiguous per-rank shards, ONE context per GPU -- so that a driver-run scaling sweep shows the
    # strong-scaling curve of the sharded batch next to the weak-scaling headline (config.extra.cfg3_strong).
    cfg3_strong = None
    if world > 1 and not strong and not args.no_extra and args.config == 2:
        cfg3_strong = cfg3_strong_line(args, dry, world, rank, local, dev, cdev, use_dist, sync)

    if rank == 0:
        fits = B_total * args.steps
        f_chol, f_fit = fit_flops(N, d, M_TEST)
        value = fits / dt
        out = {
            "metric": "GP-fits/s", "value": value, "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": dts, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: batch of independent fixed-theta GP fits, "
                                   f"N={N} d={d} M={M_TEST} kernel={'SE-ARD' if kid == 1 else 'SE-iso'} {dts}",
                       "fits_per_gpu_per_step": B if not strong else B_total / world, "fits_per_step_all_ranks": B_total,
                       "streams": args.streams, "pipeline_depth": depth, "N": N, "d": d, "M": M_TEST,
                       "single_fit_latency_ms": single_ms,
                       "single_fit_cholesky_roofline_frac": (f_chol / 1e12 / (single_ms * 1e-3) / peak) if single_ms else None,
                       "fit_tflops": value * f_fit / 1e12, "cholesky_roofline_frac": value * f_chol / 1e12 / world / peak,
                       "inputs": "resident in HBM", "ensemble": ens,
                       "ranks": world, "collective_backend": ("rccl" if backend == "nccl" else backend) if use_dist else None,
                       "collective_world": dist.get_world_size() if use_dist else None,
                       "per_rank_fits_per_s": per_rank_fits},
        }
        if dry:
            out["dry_run"] = True
            if cfg3_strong is not None:
                out["config"].setdefault("extra", {})["cfg3_strong"] = cfg3_strong
        if ablation:
            out["ablation_build"] = True   # CGP_DBG in a -DCGP_ABLATION library: NOT a measurement of the product
        if not dry:
            out["roofline"], out["kernel_ms_per_step"] = W.roofline(peak)
            if world == 1 and not args.no_extra and not strong:
                out["config"]["extra"] = extras(engine, torch, dev, local, W)
            if not args.no_cpu and world == 1:   # rank 0 at N = 1 only: other ranks would wait on the host work
                out["cpu_baseline"] = cpu_baseline(kid, X, y, Xs, th, args.cpu_sample, W.dmean, W.dvar, W.dlogml, f_fit)
                if single_ms and out["cpu_baseline"].get("value"):
                    # BASELINE configs[1] read literally is ONE fit: its latency against the one-thread CPU rate (north_star's >= 100 x
                    # is met by the batched rate, `value`; a lone fit is a chain of 16 block steps on a handful of CUs)
                    out["config"]["single_fit_vs_cpu_1thread"] = (1e3 / single_ms) / out["cpu_baseline"]["value"]
                if not args.no_extra and not strong:
                    try:
                        out["config"]["extra"]["node_cpu_baseline"] = node_cpu_leg(engine, local)
                    except Exception as e:
                        out["config"]["extra"]["node_cpu_error"] = repr(e)
                    try:
                        out["config"]["extra"]["fp32_refinement"] = refine_leg(engine, torch, dev, local)
                    except Exception as e:
                        out["config"]["extra"]["fp32_refinement_error"] = repr(e)
            if world == 1 and not args.no_extra and not strong:
                ex3 = extras_cfg3(engine, torch, dev, local, W)
                leg = ex3.pop("_cpu_leg", None)
                if leg is not None and not args.no_cpu:
                    # CPU leg for configs[2]: the C port timed on 3 of its windows, its outputs check the timed GPU outputs
                    cb3 = cpu_baseline(*leg[:5], 3, *leg[5:], context_rows=False)
                    ex3["cfg3_cpu_baseline"] = {k: cb3[k] for k in ("value", "unit", "cores", "kind", "sample")}
                    ex3["cfg3_max_rel_err_vs_oracle"] = cb3["gpu_vs_oracle_max_rel_err"]
                out["config"]["extra"].update(ex3)
            if cfg3_strong is not None:
                out["config"].setdefault("extra", {})["cfg3_strong"] = cfg3_strong
            if world == 1 and not args.no_pmc and not ablation:
                live = pmc_traffic_live(args)      # last: everything above is already measured if a pass misbehaves
                if live is not None:
                    out["roofline"]["traffic_committed"] = out["roofline"]["traffic"]
                    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = live
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cfg3_strong_line(args, dry, world, rank, local, dev, cdev, use_dist, sync, total=512, steps=20):
    """BASELINE configs[2] as written on the ranks of this job: `total` N=1024 fp32 fits per step in contiguous per-rank
    shards (sharding.shard_range), one context per GPU, steps back to back, barrier + synchronise either side, MAX over
    ranks.  Returned by every rank; rank 0 reports it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import corenav_gp_amd.synth as synth
    from corenav_gp_amd import sharding
    b0, b1 = sharding.shard_range(total, rank, world)
    B = b1 - b0
    if dry:
        def step():
            time.sleep(0.001)
    else:
        import corenav_gp_amd.engine as engine
        kid, X, y, Xs, th, dts = synth.config(3, batch=B, M=M_TEST, first=b0)
        W3 = Workload(engine, torch, dev, local, kid, X, y, Xs, th, dts, 0)   # engine defaults: a 56 ... 96-fit shard is cut into two stream groups
        step = W3.step
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:   # working clock; bursts issued back to back as the timed steps are (a mid-size shard: the
            for _ in range(8):                   # engine settles its stream groups on the pattern it sees)
                step()
            sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    dt = torch.tensor([time.perf_counter() - t0], device=cdev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if not dry:
        assert int(W3.dinfo.abs().sum().item()) == 0
    el = float(dt.item()) / steps
    return {"fits_per_s": total / el, "ms_per_step": el * 1e3, "fits_per_step_all_ranks": total, "fits_per_gpu_per_call": total / world,
            "n_gpus": world, "scaling": "strong", "pipeline_depth": 1, "streams": "engine default", "steps": steps,
            "workload": f"BASELINE configs[2] as written: {total} x N=1024 d=6 SE-ARD fp32, M={M_TEST}, sharded over {world} ranks, "
                        "one context per GPU; divide by config.extra.cfg3_fits_per_s of the --gpus 1 line for the speed-up"}


class Workload:
    """One rank's resident batch and its engine context."""

    def __init__(self, engine, torch, dev, local, kid, X, y, Xs, th, dts, streams):
        import numpy as np
        self.engine, self.torch, self.kid, self.dts = engine, torch, kid, dts
        self.X, self.y, self.Xs, self.th = X, y, Xs, th
        B, N, d = X.shape
        self.B, self.N, self.d = B, N, d
        dtype = engine.F64 if dts == "f64" else engine.F32
        tdt = torch.float64 if dts == "f64" else torch.float32
        thp = np.zeros((B, engine.MAX_THETA))
        thp[:, :th.shape[1]] = th
        self.dX = torch.from_numpy(np.ascontiguousarray(X.transpose(0, 2, 1))).to(dev, tdt)     # [B][d][N]
        self.dXs = torch.from_numpy(np.ascontiguousarray(Xs.transpose(0, 2, 1))).to(dev, tdt)   # [B][d][M]
        self.dy = torch.from_numpy(y).to(dev, tdt)
        self.dth = torch.from_numpy(thp).to(dev, torch.float64)
        self.dmean = torch.empty((B, M_TEST), device=dev, dtype=tdt)
        self.dvar = torch.empty((B, M_TEST), device=dev, dtype=tdt)
        self.dlogml = torch.empty(B, device=dev, dtype=torch.float64)
        self.dinfo = torch.zeros(B, device=dev, dtype=torch.int32)
        self.ctx = engine.Context(device=local, max_n=N, max_m=M_TEST, max_d=d, max_batch=B, dtype=dtype)
        self.ctx.set_streams(streams)
        self.stream = torch.cuda.current_stream().cuda_stream   # 0 = the legacy default stream itself

    def _call(self, nfits):
        self.ctx.fit_predict_batch_device(nfits, self.N, self.d, M_TEST, self.kid, self.dX.data_ptr(), self.dy.data_ptr(),
                                          self.dXs.data_ptr(), self.dth.data_ptr(), 0, True, self.dmean.data_ptr(),
                                          self.dvar.data_ptr(), self.dlogml.data_ptr(), self.dinfo.data_ptr(), self.stream)

    def step(self):
        self._call(self.B)

    def single_fit_latency_ms(self):
        """Latency of ONE fit of the same shape (BASELINE configs[1] reads "single GP fit"): the engine
        switches to its latency schedule for a handful of fits; synchronised per call."""
        torch = self.torch
        keep = (self.dmean[0].clone(), self.dvar[0].clone(), self.dlogml[0].clone())
        for _ in range(3):
            self._call(1)
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            self._call(1)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / 10 * 1e3
        self.dmean[0], self.dvar[0], self.dlogml[0] = keep   # the batch outputs are what the summaries / oracle check read
        return ms

    def roofline(self, peak):
        """Roofline of the dominant kernel (k_panel: trailing syrk/gemm + Gram + in-register trmm).  HIP events
        are recorded around every launch on the stream it is launched on (cgp_profile_enable) in 2 EXTRA
        steps after the timed region: per-launch events serialise the launches, so the timed steps carry
        none and the two are labelled apart."""
        # The profiled steps must run as the timed ones do: clocks up (whatever ran since the timed region --
        # single-fit launches, the gather -- lets the part idle down, and a short step is over before it is back:
        # measured 0.74 ms per fp32 launch cold against 0.62 warm = the in-kernel span, tools/launch_spans.py), and
        # the event pool created beforehand.
        self.ctx.profile_enable(True)
        self.step()
        self.torch.cuda.synchronize()
        self.ctx.profile_read()
        self.ctx.profile_enable(False)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:
            self.step()
            self.torch.cuda.synchronize()
        self.ctx.profile_enable(True)
        nprof = 2
        for _ in range(nprof):
            self.step()
        self.torch.cuda.synchronize()
        prof = self.ctx.profile_read()
        self.ctx.profile_enable(False)
        upd = prof["update"]
        achieved = upd["flops"] / (upd["ms"] * 1e-3) / 1e12 if upd["ms"] > 0 else 0.0
        traffic, tsrc = pmc_traffic(self.B, self.N, self.dts)
        fused = self.dts != "f64" or self.B < 512   # the schedule the engine takes (cgp_engine.hip: FUSED64_BELOW)
        label = ("k_panel (syrk/gemm trailing update + fused Gram + in-register trmm)" if not fused else
                 "k_panel + diagonal tile (the same launches also finish / pre-update the next diagonal tiles: flops and time of "
                 "that latency-bound factorisation are inside this figure)")
        rf = {"bound": "mfma", "kernel": label, "diag_tile_inside_launches": fused,
              "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
              "traffic": traffic, "traffic_source": tsrc,
              "from": "HIP events around every k_panel launch, 2 extra profiled steps after the timed region and "
                      "100 ms of untimed steps (not the timed steps: per-launch events serialise the launches)",
              "avg_launch_ms": upd["ms"] / max(upd["launches"], 1), "launches": upd["launches"],
              "algorithmic_flops_per_launch": upd["flops"] / max(upd["launches"], 1)}
        if self.dts == "f32" and not (self.engine.load().cgp_build_flags() & self.engine.BUILD_F32_NATIVE):
            # the fp32 tile loops run on the bf16 matrix cores (cgp_kernels_fused.hpp, bx6_compute): `peak` stays the fp32-input
            # MFMA peak -- the instruction a plain fp32 kernel has -- and the bf16 form's own ceiling is stated beside it
            rf["mfma_path"] = ("bf16x6: every fp32 product = six bf16 products (x = x0 + x1 + x2 by truncation), three K=32 bf16 MFMAs per "
                               "16x16x16 block; results at fp32 rounding level (tests/fuzz/fuzz_parity.py)")
            rf["bf16x6_bound_tflops"] = BF16_MFMA_PEAK_TFLOPS / 6.0
            rf["frac_of_bf16x6_bound"] = achieved / (BF16_MFMA_PEAK_TFLOPS / 6.0)
        return rf, {k: v["ms"] / nprof for k, v in prof.items()}


def pmc_traffic(B, N, dts):
    """HBM bytes per k_panel launch from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs of this same command, FETCH doubled for the gfx950
    half-count).  Builder-run, not measured in this process; only valid for the configuration the
    profile was taken on, otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.json")), reverse=True)
    for f in files:      # newest round first; one summary per profiled configuration
        try:
            d = json.load(open(f))
            w = d.get("_workload", {"batch": 256, "N": 2048, "dtype": "f64"})
            if (B, N, dts) != (w["batch"], w["N"], w["dtype"]):
                continue
            return (d.get("k_panel", d.get("k_update"))["hbm_bytes_per_launch"],
                    f"profiles/{os.path.basename(f)} (builder-run rocprofv3 --pmc passes of this command, not this process)")
        except Exception:
            continue
    return None, None


PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTX_", "ROCTRACER_")


def under_profiler():
    """True when this process was started under rocprofv3 / rocprofiler-sdk (its tool library is preloaded and has
    initialised the GPU before main): starting ANOTHER profiler from here would inherit the preload, every hop of
    `rocprofv3 -> env -> python3 -> app` would be an exec out of a GPU-initialised process (the pool forbids it), and
    the nested passes would pollute the outer counters.  The live PMC passes are skipped then."""
    pre = os.environ.get("LD_PRELOAD", "") + ":" + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    if "rocprofiler" in pre or "rocprof" in pre:
        return True
    return any(k.startswith(PROFILER_ENV_PREFIXES) for k in os.environ)


def pmc_traffic_live(args):
    """HBM bytes per k_panel launch of THIS command, measured during this run: two child processes
    `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --steps 1 ...` (FETCH_SIZE and WRITE_SIZE in separate
    passes, no other tracing, as MI355X_MICROARCH.md's HBM section prescribes; FETCH doubled for the gfx950
    half-count, units KB).  Children, not exec: this process keeps its GPU context.  None if rocprofv3 is missing
    or a pass fails or overruns -- the committed builder-run figure then stays in the line."""
    import csv, glob, shutil, tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("CGP_BENCH_CHILD") or under_profiler():
        return None
    tmp = tempfile.mkdtemp(prefix="cgp_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu", "--no-extra", "--no-pmc",
             "--batch", str(args.batch), "--config", str(args.config)] + (["--n", str(args.n)] if args.n else [])
    # the children get a clean environment: nothing of a profiler that may be wrapping a parent survives
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(PROFILER_ENV_PREFIXES)}
    env.update(TMPDIR="/tmp", CGP_BENCH_CHILD="1")
    rocprof = shutil.which("rocprofv3")
    mean = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            # rocprofv3 is a `#!/usr/bin/env python3` script: run it with this interpreter, no `env` hop in between
            cmd = [sys.executable, rocprof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child
            # own process group, so that an overrun ends the profiler AND the program it started (exact pgid, not a pattern)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, 9)
                except OSError:
                    pass
                pr.wait()
                return None
            if pr.returncode != 0:
                return None
            vals = []
            for f in glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_panel" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None
            mean[ctr] = sum(vals) / len(vals)
        bytes_per_launch = 2.0 * mean["FETCH_SIZE"] * 1024 + mean["WRITE_SIZE"] * 1024
        return bytes_per_launch, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command, run by this process "
                                  "after the timed region (FETCH x2 gfx950 correction, KB units)")
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def extras(engine, torch, dev, local, W):
    """Secondary lines of the same driver command (each about a second): the host-buffer (PCIe-inclusive)
    rate of the headline workload, BASELINE configs[2] (512 x N=1024 fp32), configs[3] (sliding window)
    and the batched look-ahead.  Reported under config.extra; never `value`."""
    import numpy as np
    import corenav_gp_amd.synth as synth
    ex = {}
    try:   # end to end through cgp_fit_predict_batch: pinned staging, H2D, device-side packing, D2H
        W.ctx.fit_predict_batch(W.X, W.y, W.Xs, W.th, W.kid)
        t0 = time.perf_counter()
        rc, mean, var, logml, info = W.ctx.fit_predict_batch(W.X, W.y, W.Xs, W.th, W.kid)
        el = time.perf_counter() - t0
        assert rc == 0
        ex["end_to_end_fits_per_s"] = W.B / el
        ex["end_to_end_note"] = "host fp64 buffers in, host buffers out (pinned staging + PCIe both ways + jitter check), 1 call"
        ex["end_to_end_matches_resident"] = bool(np.array_equal(logml, W.dlogml.cpu().numpy()))
    except Exception as e:
        ex["end_to_end_error"] = repr(e)
    try:
        ex.update(window_line(engine, torch, dev, local))
    except Exception as e:
        ex["window_error"] = repr(e)
    try:
        ex.update(lookahead_line(engine, local))
    except Exception as e:
        ex["lookahead_error"] = repr(e)
    try:
        ex.update(node_line(engine, local))
    except Exception as e:
        ex["node_error"] = repr(e)
    try:
        ex.update(replay_line(engine, local))
    except Exception as e:
        ex["replay_error"] = repr(e)
    return ex


def extras_cfg3(engine, torch, dev, local, W):
    """BASELINE configs[2] (512 x N=1024 fp32) on its own schedule, with its own roofline fraction and an oracle
    check (by the CPU leg, see run_rank).  Called AFTER the host-side cpu_baseline leg: measured right behind the fp64 run the part is still at
    that run's temperature / clock and reads about 8 % low (91 k against 99-101 k fits/s of a cold `--config 3`)."""
    import corenav_gp_amd.synth as synth
    ex = {}
    try:
        if W.dts == "f64":   # BASELINE configs[2] on its own schedule
            kid, X, y, Xs, th, dts = synth.config(3, batch=512, M=M_TEST)
            W3 = Workload(engine, torch, dev, local, kid, X, y, Xs, th, dts, 1)
            tw = time.perf_counter()
            while time.perf_counter() - tw < 0.15:   # the part idled through the host leg: back to its working clock first
                W3.step()
                torch.cuda.synchronize()
            n3 = 20   # (5 until round 6: the projection below moved +- 6 % between runs of one tree)
            t0 = time.perf_counter()
            for _ in range(n3):
                W3.step()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / n3
            assert int(W3.dinfo.abs().sum().item()) == 0
            rf, kms = W3.roofline(FP32_MFMA_PEAK_TFLOPS)
            ex["cfg3_fits_per_s"] = 512 / el
            ex["cfg3_ms_per_step"] = el * 1e3
            ex["cfg3_steps"] = n3
            ex["cfg3_roofline_frac"] = rf["frac"]
            # `cfg3_roofline_frac` prices the launch against the FP32-input MFMA peak (the instruction a plain fp32 kernel has).  The loop
            # itself runs on the bf16 matrix cores, eight bf16 terms per fp32 product in the triangular product and six in the tile loop:
            # against THAT pipe's ceiling (dense bf16 peak / 6, the generous bound) and against HBM the same launch reads as follows
            if "frac_of_bf16x6_bound" in rf:
                ex["cfg3_frac_of_bf16x6_bound"] = rf["frac_of_bf16x6_bound"]
                ex["cfg3_bf16x6_bound_tflops"] = rf["bf16x6_bound_tflops"]
            if rf.get("traffic") and rf.get("avg_launch_ms"):
                ex["cfg3_hbm_frac"] = rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
                ex["cfg3_hbm_note"] = ("PMC bytes per k_panel launch (" + str(rf.get("traffic_source")) + ") / the launch's duration in this run / 8 TB/s: "
                                       "neither the matrix cores nor HBM is saturated")
            ex["cfg3_kernel_ms_per_step"] = kms
            ex["cfg3_workload"] = "BASELINE configs[2]: 512 x N=1024 d=6 SE-ARD fp32, M=599, one GPU's view of the sweep"
            ex["cfg3_strong"] = {"fits_per_s": 512 / el, "ms_per_step": el * 1e3, "fits_per_step_all_ranks": 512, "fits_per_gpu_per_call": 512,
                                 "n_gpus": 1, "scaling": "strong", "pipeline_depth": 1, "steps": n3,
                                 "workload": "BASELINE configs[2] as written on ONE rank (the denominator of the strong-scaling curve; "
                                             "`bench.py --gpus N` reports the same key over N ranks)"}
            # configs[2] AS WRITTEN shards the 512 fits over 8 GPUs: 64 fits per GPU and call.  The same engine on the
            # first 64 windows (its mid-size schedule), and what 8 such GPUs would make of the 1-GPU rate above -- a
            # projection from this GPU's two rates (no collective on the data path), not a measurement of 8 GPUs.
            W64 = Workload(engine, torch, dev, local, kid, X[:64], y[:64], Xs[:64], th[:64], dts, 0)   # default settings: the engine picks the stream groups
            tw = time.perf_counter()
            while time.perf_counter() - tw < 0.1:   # bursts issued back to back, as the timed calls are: the engine settles its stream groups on this pattern
                for _ in range(8):
                    W64.step()
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                W64.step()
            torch.cuda.synchronize()
            el64 = (time.perf_counter() - t0) / 20
            assert int(W64.dinfo.abs().sum().item()) == 0
            ex["cfg3_fits_per_s_at_64"] = 64 / el64
            ex["cfg3_ms_per_call_at_64"] = el64 * 1e3
            ex["cfg3_strong_scaling_projection_8gpu"] = 8 * (64 / el64) / (512 / el)
            ex["cfg3_strong_scaling_note"] = ("8 x (64-fit call rate) / (512-fit call rate) on this one GPU, ONE context, default settings (the engine "
                                              "cuts a 56 ... 96-fit fp32 call into two stream groups by itself since round 5); measure it with "
                                              "`bench.py --scaling strong --config 3 --gpus 8`")
            # the same call forced into ONE group (cgp_set_streams(1)): what every round before 5 reported under this key
            W64.ctx.set_streams(1)
            for _ in range(10):
                W64.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                W64.step()
            torch.cuda.synchronize()
            el64g = (time.perf_counter() - t0) / 20
            W64.ctx.set_streams(0)
            assert int(W64.dinfo.abs().sum().item()) == 0
            ex["cfg3_ms_per_call_at_64_one_group"] = el64g * 1e3
            try:
                ex["cgp_sweep"] = sweep_line(engine, torch, dev, local, W64, kid, el64)
            except Exception as e:
                ex["cgp_sweep_error"] = repr(e)
            # the same 64-fit calls dealt over TWO contexts on two HIP streams (--pipeline 2):
            # successive calls overlap, the chain-bound early block steps of one beside the MFMA-bound late ones of the other
            W64b = Workload(engine, torch, dev, local, kid, X[:64], y[:64], Xs[:64], th[:64], dts, 1)
            lanes = [W64, W64b]
            for ln in lanes:
                ln.torch_stream = torch.cuda.Stream(dev)
                ln.stream = ln.torch_stream.cuda_stream
            torch.cuda.synchronize()
            for i in range(8):
                lanes[i % 2].step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                lanes[i % 2].step()
            torch.cuda.synchronize()
            el64p = (time.perf_counter() - t0) / 40
            assert int(W64.dinfo.abs().sum().item()) == 0 and int(W64b.dinfo.abs().sum().item()) == 0
            assert torch.equal(W64.dmean, W64b.dmean) and torch.equal(W64.dlogml, W64b.dlogml)
            ex["cfg3_fits_per_s_at_64_pipelined2"] = 64 / el64p
            ex["cfg3_strong_scaling_projection_8gpu_pipelined2"] = 8 * (64 / el64p) / (512 / el)
            ex["cfg3_pipelined2_note"] = ("two successive sweeps' 64-fit calls overlapped on two contexts per GPU: throughput of REPEATED sweeps "
                                          "(cross-sweep overlap), not the speed-up of one 512-fit sweep -- that is cfg3_strong_scaling_projection_8gpu")
            del W64, W64b
            # for the CPU leg (the only place of this program that may run the oracle): inputs and the timed outputs
            ex["_cpu_leg"] = (kid, X, y, Xs, th, W3.dmean, W3.dvar, W3.dlogml, fit_flops(1024, 6, M_TEST)[1])
            del W3
    except Exception as e:
        ex["cfg3_error"] = repr(e)
    return ex


def sweep_line(engine, torch, dev, local, W64, kid, el_ctx):
    """The C-ABI multi-device entry (cgp_sweep_fit_predict_device: what a C++ ROS host calls; per-shard device pointers, work
    enqueued by persistent per-device worker threads, no copy and no synchronisation inside the call) on the 64-fit
    configs[2] shard, next to the plain context's figure: devices = [local] (must be the same call), [local, local] (two
    contexts of 32 fits on one GPU: the one-GPU stand-in of two devices), and -- when this process sees more than one
    GPU -- ALL visible devices from this ONE process, 64 fits each."""
    def timed(sw, B, ptrs, reps=20):
        def sync():
            sw.synchronize()                    # the contexts' own streams
            for i in range(torch.cuda.device_count()):
                torch.cuda.synchronize(i)       # ... and whatever stream the caller passed
        for _ in range(24):                     # back to back, as the timed calls: the engine settles its stream groups on this pattern
            sw.fit_predict_device(B, W64.N, W64.d, M_TEST, kid, *ptrs)
        sync()
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.1:   # creating the sweep's contexts let the part idle: back to its working clock first
            for _ in range(4):
                sw.fit_predict_device(B, W64.N, W64.d, M_TEST, kid, *ptrs)
            sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            sw.fit_predict_device(B, W64.N, W64.d, M_TEST, kid, *ptrs)
        sync()
        return (time.perf_counter() - t0) / reps
    dtype = engine.F64 if W64.dts == "f64" else engine.F32
    out = {"workload": "64 x N=1024 d=6 fp32 (BASELINE configs[2], one GPU's shard), device-resident, cgp_sweep_fit_predict_device",
           "context_ms_per_call": el_ctx * 1e3}
    whole = ([W64.dX.data_ptr()], [W64.dy.data_ptr()], [W64.dXs.data_ptr()], [W64.dth.data_ptr()], None, True,
             [W64.dmean.data_ptr()], [W64.dvar.data_ptr()], [W64.dlogml.data_ptr()], [W64.dinfo.data_ptr()])
    sw = engine.Sweep([local], W64.N, M_TEST, W64.d, 64, dtype)
    el1 = timed(sw, 64, (*whole, [W64.stream]))     # on the stream the context's figure was measured on
    out["devices_1_ms_per_call"] = el1 * 1e3
    out["devices_1_over_context"] = el1 / el_ctx
    el1o = timed(sw, 64, whole)                     # hip_streams = NULL: the sweep's contexts' own streams
    out["devices_1_own_stream_ms_per_call"] = el1o * 1e3
    out["stream_note"] = ("whether the two stream groups of a 56 ... 96-fit fp32 call overlap depends on the runtime's stream -> hardware-queue "
                          "mapping (process history); the engine times both forms per caller stream and keeps the faster, so a stream "
                          "on which they cannot overlap runs the call as one group (cfg3_ms_per_call_at_64_one_group)")
    sw.close()
    h = 32
    halves = tuple([t[:h].data_ptr(), t[h:].data_ptr()] for t in (W64.dX, W64.dy, W64.dXs, W64.dth))
    outs = tuple([t[:h].data_ptr(), t[h:].data_ptr()] for t in (W64.dmean, W64.dvar, W64.dlogml, W64.dinfo))
    sw = engine.Sweep([local, local], W64.N, M_TEST, W64.d, 64, dtype)
    el2 = timed(sw, 64, (*halves, None, True, *outs))
    out["devices_same_gpu_twice_ms_per_call"] = el2 * 1e3
    sw.close()
    assert int(W64.dinfo.abs().sum().item()) == 0
    ndev = torch.cuda.device_count()
    if ndev > 1:   # one process, every visible GPU: 64 fits per device (weak), inputs replicated device by device
        bufs = []
        for i in range(ndev):
            di = torch.device("cuda", i)
            bufs.append({k: getattr(W64, k).to(di) for k in ("dX", "dy", "dXs", "dth", "dmean", "dvar", "dlogml", "dinfo")})
        for i in range(ndev):
            torch.cuda.synchronize(i)
        ptr = lambda k: [b[k].data_ptr() for b in bufs]
        sw = engine.Sweep(list(range(ndev)), W64.N, M_TEST, W64.d, 64 * ndev, dtype)
        eln = timed(sw, 64 * ndev, (ptr("dX"), ptr("dy"), ptr("dXs"), ptr("dth"), None, True, ptr("dmean"), ptr("dvar"), ptr("dlogml"), ptr("dinfo")))
        out["all_visible_devices"] = {"devices": ndev, "fits_per_call": 64 * ndev, "ms_per_call": eln * 1e3,
                                      "fits_per_s": 64 * ndev / eln, "speedup_over_one_device": (64 * ndev / eln) / (64 / el1)}
        assert all(int(b["dinfo"].abs().sum().item()) == 0 for b in bufs)
        sw.close()
        torch.cuda.set_device(local)
    return out


def replay_line(engine, local, n_traj=64, ticks=600):
    """The closed loop of BASELINE configs[4]'s stand-in (corenav_gp_amd/replay.py: synthetic rover -> recorder -> GP engine ->
    batched look-ahead -> stop command -> zero updates -> next SetStopping answer) for a 64-trajectory Monte-Carlo
    ensemble on one GPU: wall-clock per simulated second and windows fitted."""
    from corenav_gp_amd import replay
    ens = replay.ClosedLoopEnsemble(n_traj=n_traj, device=local)
    t0 = time.perf_counter()
    npub = ens.run(ticks)
    el = time.perf_counter() - t0
    return {"replay": {"trajectories": n_traj, "ticks": ticks, "simulated_s": ticks * replay.DT_ODO, "wall_s": el,
                       "windows_fitted": int(npub), "stops": int(sum(t.stops for t in ens.traj)),
                       "trajectory_seconds_per_wall_second": n_traj * ticks * replay.DT_ODO / el,
                       "note": "host-side rover / recorder / covariance model in Python dominates the wall clock; the GPU work is "
                               "one batched fit + predict and one batched look-ahead per published tick"}}


def window_line(engine, torch, dev, local, W=1024, N=512, d=3, T=200):
    """BASELINE configs[3]: sliding-window GP, N = 512 ring, one rank-1 up/downdate per tick.  `window_hbm_frac` prices
    the ALGORITHMIC bytes of a tick -- the factor read and written once, n^2/2 * 8 B * 2 -- against the HBM
    peak; since round 3 steady-state ticks go two per pass over the factor, so the traffic that actually reaches HBM is
    about half that figure (`window_traffic_note`)."""
    import numpy as np
    rng = np.random.default_rng(20264)
    t = np.arange(11, 11 + N + T, dtype=np.float64)
    X = np.empty((W, len(t), d))
    X[:, :, 0] = (t - t.mean()) / t.std()
    X[:, :, 1:] = rng.normal(size=(W, len(t), d - 1))
    y = 0.1 * np.sin(2 * np.pi * t / 40.0)[None] + rng.normal(0, 0.03, (W, len(t)))
    theta = np.concatenate([[0.02], np.linspace(0.8, 1.6, d), [1e-3]])
    ctx = engine.Context(device=local, max_n=8, max_m=8, max_d=d)
    ctx.window_init(W, N, d, 1, theta)
    dX, dy = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)

    def push(a, b):
        xs, ys = dX[:, a:b].contiguous(), dy[:, a:b].contiguous()
        out = torch.empty((3, W, b - a), device=dev, dtype=torch.float64)
        ctx.window_push_device(b - a, xs.data_ptr(), ys.data_ptr(), True, out[0].data_ptr(), out[1].data_ptr(),
                               out[2].data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out
    push(0, N)                       # fill the windows (warm-up, not timed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    push(N, N + T)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert ctx.window_state(0)[1] == 0
    gbps = W * T * (N * N / 2 * 8 * 2) / el / 1e9
    # the per-tick host entry (cgp_window_push, T = 1, ONE window, host buffers in and out: what a node that receives one
    # sample per IMU tick calls): pinned staging kept in the context, one H2D, the launch, two D2H, one synchronisation
    c1 = engine.Context(device=local, max_n=8, max_m=8, max_d=d)
    c1.window_init(1, N, d, 1, theta)
    c1.window_push(X[:1, :N], y[:1, :N])                 # fill
    for i in range(5):
        c1.window_push(X[:1, N + i:N + i + 1], y[:1, N + i:N + i + 1])
    nt = 50
    t1 = time.perf_counter()
    for i in range(5, 5 + nt):
        c1.window_push(X[:1, N + i:N + i + 1], y[:1, N + i:N + i + 1])
    host_tick_us = (time.perf_counter() - t1) / nt * 1e6
    # counter traffic of the same command (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, profiles/r06_window_pmc_summary.json, the round's final
    # kernels: k_window_multi<4> takes the steady-state ticks): 2 x FETCH_SIZE KB + WRITE_SIZE KB = 143.8 GB for the 429.5 GB algorithmic of
    # 200 ticks x 1024 windows
    kCounterOverAlgorithmic = 143.8 / 429.5
    tpp = 4 if (W >= 512 and N >= 64) else 2      # steady-state ticks per pass over the factor (k_window_multi<4> from 512 windows, k_window_pairs below)
    return {"window_ticks_per_s": W * T / el, "window_hbm_frac": gbps / tpp / HBM_PEAK_GBPS, "window_ticks_per_pass": tpp,
            "window_hbm_frac_one_pass_per_tick_equivalent": gbps / HBM_PEAK_GBPS,
            "window_hbm_frac_counter_traffic_projected": gbps / HBM_PEAK_GBPS * kCounterOverAlgorithmic, "window_host_tick_us": host_tick_us,
            "window_counter_over_algorithmic": {"ratio": kCounterOverAlgorithmic, "from": "profiles/r06_window_pmc_summary.json (the final round-6 "
                                                "window kernels), NOT measured in this run"},
            "window_traffic_note": "a pass over the factor reads and writes it once (n^2/2 x 8 B x 2) and advances window_ticks_per_pass ticks: "
                                   "window_hbm_frac = those bytes x passes/s / 8 TB/s, measured in this run; _one_pass_per_tick_equivalent = the same rate priced "
                                   "at one pass per tick (what rounds 1-5 reported; above 1 now); the PMC counters saw 0.33 of the per-tick bytes: "
                                   "window_hbm_frac_counter_traffic_projected = this run's rate x that committed ratio, a projection (tools/pmc_window.sh "
                                   "re-measures the ratio)",
            "window_workload": f"BASELINE configs[3]: {W} windows x N={N} d={d} fp64, {T} ticks, algorithmic {gbps:.0f} GB/s of {HBM_PEAK_GBPS:.0f}"}


def lookahead_line(engine, local, T=4096):
    """Batched stop-time look-ahead (GpPredictor::GPCallBack arithmetic, one wavefront per trajectory)."""
    import numpy as np
    import corenav_gp_amd.synth as synth
    g = np.load(os.path.join(ROOT, "tests", "golden", "lookahead_restated.npz"))
    states = [synth.filter_state(5000 + k) for k in range(64)]
    P, Q, STM, Hv, pos = (np.stack([states[k % 64][j] for k in range(T)]) for j in range(5))
    means, sigmas = np.tile(g["mean"], (T, 1)), np.tile(g["sigma"], (T, 1))
    P[::3] *= 1e-4   # a third of the ensemble never crosses the threshold: full 599 x 5 propagation steps
    Q[::3] *= 1e-4
    ctx = engine.Context(device=local, max_n=8, max_m=8, max_d=1)
    ctx.predict_stop_batch(means, sigmas, P, Q, STM, Hv, pos, 10.0, 10.0)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.predict_stop_batch(means, sigmas, P, Q, STM, Hv, pos, 10.0, 10.0)
    el = (time.perf_counter() - t0) / 3
    return {"lookahead_traj_per_s": T / el, "lookahead_workload": f"{T} trajectories x 599 x 5 steps, host buffers (PCIe copies included)"}


def node_line(engine, local):
    """The reference node's own work item (gp_slip_node.py:16-63): ONE GP_Input window of its recorded slip series
    (149 ticks, 90 % kept, 599 predicted ticks, RBF x Brownian), host buffers in and out, with the fixed theta of the
    fixture and with the reference's m.optimize() from theta = ones."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "slipval_window_rbfbrownian.npz"))
    t, s, th = g["time_array"], g["slip_array"], g["theta"]
    ctx = engine.Context(device=local, max_n=256, max_m=1024, max_d=1, max_batch=1)
    for _ in range(3):
        ctx.slip_node_callback(t, s, th)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        ctx.slip_node_callback(t, s, th)
        ts.append(time.perf_counter() - t0)
    ctx.slip_node_callback_opt(t, s, np.ones(4))
    to = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.slip_node_callback_opt(t, s, np.ones(4))
        to.append(time.perf_counter() - t0)
    # the same work item for an ensemble: 256 windows of the series (shifted starts) in one batched call, fixed theta
    W, n = 256, int(0.9 * len(t))
    X = np.stack([t[:n] + k for k in range(W)])[:, :, None]
    y = np.stack([np.roll(s, k)[:n] for k in range(W)])
    Xs = np.stack([X[k, -1, 0] + 1 + np.arange(599.0) for k in range(W)])[:, :, None]
    thw = np.tile(th, (W, 1))
    bctx = engine.Context(device=local, max_n=n, max_m=599, max_d=1, max_batch=W)
    for _ in range(2):
        rc, _, _, _, info = bctx.fit_predict_batch(X, y, Xs, thw, engine.KERNEL_RBF_BROWNIAN)
    assert rc == 0 and not info.any()
    tb = []
    for _ in range(5):
        t0 = time.perf_counter()
        bctx.fit_predict_batch(X, y, Xs, thw, engine.KERNEL_RBF_BROWNIAN)
        tb.append(time.perf_counter() - t0)
    return {"node_callback_us": 1e6 * float(np.median(ts)), "node_callback_opt_ms": 1e3 * float(np.median(to)),
            "node_windows_per_s": W / float(np.median(tb)),
            "node_windows_workload": f"{W} such windows ({n} kept ticks, 599 predictions each) in one cgp_fit_predict_batch call, host buffers (PCIe copies included)",
            "node_workload": f"one GP_Input window of the reference's slip series ({len(t)} ticks, 599 predictions, RBF x Brownian, fp64), "
                             "host buffers, per callback: fixed theta / with m.optimize() from theta = ones"}


def node_cpu_leg(engine, local):
    """The reference node's own work item on the host (test infrastructure: oracle/gp_oracle.py, numpy + scipy LAPACK and
    scipy's L-BFGS-B held to ONE thread), beside the engine's callback on the same window; the oracle's outputs check the
    engine's.  gp_slip_node.py:16-63."""
    import numpy as np
    from oracle import gp_oracle as go
    g = np.load(os.path.join(ROOT, "tests", "golden", "slipval_window_rbfbrownian.npz"))
    t, s, th = g["time_array"], g["slip_array"], g["theta"]
    ctx = engine.Context(device=local, max_n=256, max_m=1024, max_d=1, max_batch=1)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    try:
        go.slip_node_callback(t, s, th)
        t0 = time.perf_counter()
        for _ in range(5):
            omean, osigma = go.slip_node_callback(t, s, th)
        fixed = (time.perf_counter() - t0) / 5
        _, _, Xtr, Ytr = go.slip_node_split(t, s)
        ytr = Ytr[:, 0]
        t0 = time.perf_counter()
        tho, ologml, nev = go.optimize(go.KERNEL_RBF_BROWNIAN, Xtr, ytr)
        om2, os2 = go.slip_node_callback(t, s, tho)
        opt = time.perf_counter() - t0
    finally:
        if limit is not None:
            limit.restore_original_limits()
    mean, sigma = ctx.slip_node_callback(t, s, th)
    m2, s2, th2 = ctx.slip_node_callback_opt(t, s, np.ones(4))
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - b)) / max(np.max(np.abs(b)), 1e-300))
    # the two optimisers stop at their own resolution: compare the objective reached, and the fixed-theta outputs exactly
    f_gpu = go.nll_and_grad(go.KERNEL_RBF_BROWNIAN, th2, Xtr, ytr)[0]
    return {"fixed_theta_ms": fixed * 1e3, "with_optimize_ms": opt * 1e3, "optimizer_evaluations": int(nev), "cores": 1, "kind": "port",
            "sample": "one GP_Input window of the reference's slip series (149 ticks, 599 predictions, RBF x Brownian), oracle/gp_oracle.py on one host thread",
            "gpu_vs_oracle_max_rel_err_fixed_theta": max(rel(mean, omean), rel(sigma, osigma)),
            "nll_at_gpu_optimum_minus_nll_at_oracle_optimum": float(f_gpu - (-ologml))}


def refine_leg(engine, torch, dev, local):
    """fp32 accuracy on the windows round 5's sweep flagged (dense, one input dimension, N = 1000): the predictive mean of eight
    fits of a 40-fit call against the oracle (test infrastructure, this CPU leg only) without and with the engine's default
    refinement of alpha against a double-precision residual (cgp_set_refine, csrc/cgp_refine.hpp), and what the call costs either way."""
    import numpy as np
    import corenav_gp_amd.synth as synth
    from oracle import gp_oracle as go
    N, M, d, B, seed = 1000, 5, 1, 40, 12345
    Xl, yl, Xsl, thl = [], [], [], []
    for b in range(B):
        X, y, Xs = synth.window(N, d, M, seed + b)
        Xl.append(X); yl.append(y); Xsl.append(Xs); thl.append(synth.theta_for(1, d, y, None))
    X, y, Xs, th = np.stack(Xl), np.stack(yl), np.stack(Xsl), np.stack(thl)
    ctx = engine.Context(device=local, max_n=N, max_m=M, max_d=d, max_batch=B, dtype=engine.F32)
    out, ms = {}, {}
    for mode in (0, -1):
        ctx.set_refine(mode)
        ctx.fit_predict_batch(X, y, Xs, th, 1)
        t0 = time.perf_counter()
        for _ in range(3):
            rc, mean, var, logml, info = ctx.fit_predict_batch(X, y, Xs, th, 1)
        ms[mode] = (time.perf_counter() - t0) / 3 * 1e3
        assert rc == 0
        out[mode] = (mean, var, logml)
    err = {0: [], -1: [], "var": [], "logml": []}
    for b in range(0, B, 5):
        f = go.fit(1, th[b], X[b], y[b])
        omu, ovar = go.predict(f, Xs[b])
        sc = max(float(np.max(np.abs(omu))), 0.1 * float(np.max(np.abs(y[b]))))
        for mode in (0, -1):
            err[mode].append(float(np.max(np.abs(out[mode][0][b] - omu))) / sc)
        err["var"].append(float(np.max(np.abs(out[-1][1][b] - ovar) / np.abs(ovar))))
        err["logml"].append(abs(out[-1][2][b] - f.logml) / abs(f.logml))
    return {"workload": f"{B} x N={N} d={d} SE-ARD fp32, M={M}: dense one-dimensional windows (round 5's open accuracy item)",
            "mean_err_vs_oracle_unrefined_max": max(err[0]), "mean_err_vs_oracle_default_max": max(err[-1]),
            "variance_err_vs_oracle_max": max(err["var"]), "logml_err_vs_oracle_max": max(err["logml"]), "fits_compared": len(err[0]),
            "host_call_ms_unrefined": ms[0], "host_call_ms_default": ms[-1],
            "note": "default = cgp_set_refine(-1): one correction step for every fit of a window with d <= 3 (and for the dense fits beyond); "
                    "variance and logML come from the single-precision factor either way"}


def host_description():
    model, gov = "unknown", "unavailable"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        gov = open("/sys/devices/system/cpu/cpu0/cpufreq/scaling_governor").read().strip()
    except Exception:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    quota = "unavailable"      # cgroup v2 CPU quota of this container: "<quota_us|max> <period_us>"
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
        q, per = quota.split()
        if q != "max":
            usable = max(1, min(usable, int(float(q) / float(per))))
    except Exception:
        pass
    return {"cpu_model": model, "governor": gov, "logical_cpus": os.cpu_count(), "usable_cpus": usable, "cgroup_cpu_max": quota}


def cpu_baseline(kid, X, y, Xs, th, nsample, dmean, dvar, dlogml, f_fit, context_rows=True):
    """The oracle ("port" of the reference arithmetic) timed on this box's host cores, ONE thread like the
    reference's catkin build, on a bounded sample of the same windows (warm-ups, then the MEDIAN of
    the per-fit times).  Three single-thread implementations of the same restatement are timed -- oracle/gp_oracle.c
    (plain C, gcc -O3 -march=native here), oracle/gp_oracle.py on ONE LAPACK thread, and oracle/gp_oracle_lapack.c (C on
    dpotrf / dpotrs / dtrsm of the bundled OpenBLAS, one thread, vectorised Gram loop: the Eigen::LLT-class row) -- and
    the FASTEST is `value`: the honest single-thread denominator.  The C port's
    outputs also check the timed GPU outputs.  Context row (not the baseline): the C port with one window per host
    thread over every CPU the container may use (cgroup quota stated)."""
    import ctypes
    import statistics
    import tempfile
    import numpy as np
    # compiled HERE (-march=native of the box that runs the baseline), not shipped from the build container
    so = os.path.join(tempfile.mkdtemp(prefix="cgp_oracle_"), "libgp_oracle.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "gp_oracle.c"), "-lm"])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    nwin, N, d = X.shape
    M = Xs.shape[1]
    host = host_description()

    def run(b):
        mean, var, logml, jit = np.zeros(M), np.zeros(M), np.zeros(1), np.zeros(1)
        p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
        Xb, yb, Xsb, thb = (np.ascontiguousarray(a[b % nwin], dtype=np.float64) for a in (X, y, Xs, th))
        t0 = time.perf_counter()
        rc = lib.oracle_fit_predict(kid, p(thb), N, d, p(Xb), p(yb), M, p(Xsb), 1, p(mean), p(var), p(logml), None,
                                    None, p(jit))
        return time.perf_counter() - t0, rc, mean, var, float(logml[0])

    t_first = run(0)[0]
    warm = 5 if t_first < 0.05 else 1
    for i in range(warm - 1):
        run(i + 1)
    # median of >= 20 reps (>= 3 for N = 2048 single-thread), about 8 s of host work
    reps = nsample if nsample > 0 else int(min(20, max(3, 8.0 / max(t_first, 1e-4))))
    gm, gv, gl = dmean.cpu().numpy().astype(np.float64), dvar.cpu().numpy().astype(np.float64), dlogml.cpu().numpy()
    times, worst = [], 0.0
    for b in range(reps):
        el, rc, mean, var, logml = run(b)
        assert rc == 0
        times.append(el)
        if b < nwin:   # parity check on every timed output
            worst = max(worst, float(np.max(np.abs(gm[b] - mean)) / np.max(np.abs(mean))),
                        float(np.max(np.abs(gv[b] - var) / var)), abs(gl[b] - logml) / abs(logml))
    med = statistics.median(times)
    tol = 1e-6 if str(dmean.dtype).endswith("float64") else 1e-3   # north_star parity bar
    assert worst < tol, f"timed GPU outputs differ from the oracle: max rel err {worst:.3e} >= {tol}"
    rows = {"c_port_1_thread": {"value": 1.0 / med, "unit": "fits/s", "threads": 1,
                                "sample": f"median of {reps} of the step's windows after {warm} warm-up(s) through oracle/gp_oracle.c "
                                          f"(gcc -O3 -march=native), {sum(times):.1f} s; {f_fit / med / 1e9:.2f} GFLOP/s"}}
    # the same restatement on one LAPACK thread (numpy/scipy: dpotrf, dtrtrs)
    try:
        from threadpoolctl import threadpool_limits
        from oracle import gp_oracle as go
        nl = 3 if context_rows else 2
        with threadpool_limits(limits=1):
            go.predict(go.fit(kid, th[0], X[0], y[0]), Xs[0])   # warm-up
            ts = []
            for b in range(nl):
                t1 = time.perf_counter()
                go.predict(go.fit(kid, th[b % nwin], X[b % nwin], y[b % nwin]), Xs[b % nwin])
                ts.append(time.perf_counter() - t1)
        m1 = statistics.median(ts)
        rows["lapack_1_thread"] = {"value": 1.0 / m1, "unit": "fits/s", "threads": 1,
                                   "sample": f"median of {nl} windows after 1 warm-up through oracle/gp_oracle.py, numpy/scipy LAPACK "
                                             f"limited to ONE thread (threadpoolctl); {f_fit / m1 / 1e9:.2f} GFLOP/s"}
    except Exception as e:
        rows["lapack_1_thread"] = {"error": repr(e)}
    lapack_run = None
    # the same steps in C on LAPACK / BLAS (dpotrf, dpotrs, dtrsm, dgemv of scipy's bundled OpenBLAS, dlopen'ed and held to ONE
    # thread) with a vectorised Gram loop: the "Eigen::LLT-class" row -- no numpy temporaries in the denominator
    try:
        import glob
        import scipy
        blas = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so")))
        so2 = os.path.join(os.path.dirname(so), "libgp_oracle_lapack.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fno-math-errno", "-fPIC", "-std=gnu11", "-shared", "-o", so2,
                               os.path.join(ROOT, "oracle", "gp_oracle_lapack.c"), "-lm", "-ldl"])
        lib2 = ctypes.CDLL(so2)
        lib2.oracle_lapack_init.argtypes = [ctypes.c_char_p]
        if not blas or lib2.oracle_lapack_init(blas[0].encode()) != 0:
            raise RuntimeError("no bundled OpenBLAS to dlopen")

        def run2(b):
            mean, var, logml, jit, tp = np.zeros(M), np.zeros(M), np.zeros(1), np.zeros(1), np.zeros(1)
            p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
            Xb, yb, Xsb, thb = (np.ascontiguousarray(a[b % nwin], dtype=np.float64) for a in (X, y, Xs, th))
            t0 = time.perf_counter()
            rc = lib2.oracle_fit_predict_lapack(kid, p(thb), N, d, p(Xb), p(yb), M, p(Xsb), 1, p(mean), p(var), p(logml), None,
                                                p(jit), p(tp))
            return time.perf_counter() - t0, rc, mean, var, float(logml[0]), float(tp[0])

        run2(0)
        n2 = max(3, min(20, int(4.0 / max(run2(0)[0], 1e-4))))
        res = [run2(b) for b in range(n2)]
        assert not any(r[1] for r in res)
        w2 = max(max(float(np.max(np.abs(gm[b] - r[2])) / np.max(np.abs(r[2]))), float(np.max(np.abs(gv[b] - r[3]) / r[3])),
                     abs(gl[b] - r[4]) / abs(r[4])) for b, r in enumerate(res) if b < nwin)
        assert w2 < tol, f"timed GPU outputs differ from the LAPACK oracle: {w2:.3e}"
        m2 = statistics.median(r[0] for r in res)
        tpo = statistics.median(r[5] for r in res)
        lapack_run = run2
        rows["c_lapack_1_thread"] = {"value": 1.0 / m2, "unit": "fits/s", "threads": 1,
                                     "dpotrf_gflops": N ** 3 / 3.0 / tpo / 1e9,
                                     "sample": f"median of {n2} of the step's windows after 2 warm-ups through oracle/gp_oracle_lapack.c (gcc -O3 "
                                               f"-march=native; dpotrf / dpotrs / dtrsm / dgemv of scipy's bundled OpenBLAS on ONE thread, "
                                               f"vectorised Gram loop); {f_fit / m2 / 1e9:.2f} GFLOP/s over the fit, dpotrf alone "
                                               f"{N ** 3 / 3.0 / tpo / 1e9:.1f} GFLOP/s"}
    except Exception as e:
        rows["c_lapack_1_thread"] = {"error": repr(e)}
    best = max((k for k in rows if "value" in rows[k]), key=lambda k: rows[k]["value"])
    base = {"value": rows[best]["value"], "unit": "fits/s", "cores": 1, "kind": "port",
            "sample": f"the fastest of the single-thread runs of the oracle: {best} -- " + rows[best]["sample"],
            "single_thread_rows": rows, "host": host, "gpu_vs_oracle_max_rel_err": worst}
    if not context_rows:
        return base
    # context row: one window per host thread (ctypes releases the GIL), every CPU the container may use
    try:
        from concurrent.futures import ThreadPoolExecutor
        ncore = max(1, host["usable_cpus"])   # affinity mask capped by the cgroup quota (host["cgroup_cpu_max"])
        nw = max(256, 2 * ncore) if med * max(256, 2 * ncore) / ncore < 20.0 else 2 * ncore   # keep the row under ~20 s on small hosts
        with ThreadPoolExecutor(max_workers=ncore) as ex:
            list(ex.map(lambda i: run(i)[1], range(ncore)))   # warm-up round
            t2 = time.perf_counter()
            rcs = list(ex.map(lambda i: run(i)[1], range(nw)))
            el2 = time.perf_counter() - t2
        assert not any(rcs)
        base["port_threads_over_batch"] = {"value": nw / el2, "unit": "fits/s", "threads": ncore,
                                           "kind": "C port, one window per host thread, all usable CPUs of the container",
                                           "sample": f"{nw} windows after a {ncore}-window warm-up, {el2:.1f} s"}
    except Exception as e:
        base["port_threads_over_batch"] = {"error": repr(e)}
    # the same with the LAPACK row: what every core of the container makes of the sweep (each call single-threaded, one window
    # per host thread) -- the strongest CPU figure this host offers, for scale; not the reference's single-threaded build
    if lapack_run is not None:
        try:
            from concurrent.futures import ThreadPoolExecutor
            ncore = max(1, host["usable_cpus"])
            nw = 8 * ncore
            with ThreadPoolExecutor(max_workers=ncore) as ex:
                list(ex.map(lambda i: lapack_run(i)[1], range(ncore)))
                t2 = time.perf_counter()
                rcs = list(ex.map(lambda i: lapack_run(i)[1], range(nw)))
                el2 = time.perf_counter() - t2
            assert not any(rcs)
            base["lapack_threads_over_batch"] = {"value": nw / el2, "unit": "fits/s", "threads": ncore,
                                                 "kind": "C + OpenBLAS (one thread per call), one window per host thread, all usable CPUs of the container",
                                                 "sample": f"{nw} windows after a {ncore}-window warm-up, {el2:.1f} s"}
        except Exception as e:
            base["lapack_threads_over_batch"] = {"error": repr(e)}
    return base


if __name__ == "__main__":
    main()
