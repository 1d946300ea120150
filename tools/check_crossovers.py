#!/usr/bin/env python3
"""Guards the engine's schedule table (csrc/cgp_engine.hip: lat_fits_by_steps, mid_fits, FUSED64_BELOW): the crossovers were
measured on single boxes while box-to-box spread is several per cent, so nothing but a re-measurement says the shipped
choice is still the faster one.  For points either side of every switch this times the SAME call three ways through the
measurement library (libcorenav_gp_ab.so, same kernels; its environment overrides force a schedule) -- the shipped choice
and the switch forced either way -- each in a fresh process, and fails (exit 1) when the shipped choice is more than
TOL slower than the best forced alternative.

usage: python3 tools/check_crossovers.py [--tol 0.07] [--quick]     (prints a table; keep a copy under profiles/)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "corenav_gp_amd", "libcorenav_gp_ab.so")

# (switch, dtype, N, batch, {name: env}) -- the alternatives of the switch at that point
LAT = {"latency": {"CGP_LAT_FITS": "64"}, "throughput": {"CGP_LAT_FITS": "0"}}
MID = {"mid-size": {"CGP_MID_FITS": "512", "CGP_LAT_FITS": "0"}, "full": {"CGP_MID_FITS": "0", "CGP_LAT_FITS": "0"}}
FUS = {"diag inside": {"CGP_SCHED": "fuseddiag"}, "diag launches": {"CGP_SCHED": "splitdiag"}}
POINTS = [
    ("latency | throughput", "f64", 2048, 11, LAT), ("latency | throughput", "f64", 2048, 12, LAT),
    ("latency | throughput", "f64", 1024, 18, LAT), ("latency | throughput", "f64", 1024, 19, LAT),
    ("latency | throughput", "f64", 256, 48, LAT), ("latency | throughput", "f64", 256, 49, LAT),
    ("latency | throughput", "f32", 1024, 20, LAT), ("latency | throughput", "f32", 1024, 21, LAT),
    ("latency | throughput", "f32", 512, 24, LAT), ("latency | throughput", "f32", 512, 25, LAT),
    ("mid-size | full", "f32", 1024, 96, MID), ("mid-size | full", "f32", 1024, 97, MID),
    ("mid-size | full", "f64", 2048, 96, MID), ("mid-size | full", "f64", 2048, 97, MID),
    ("mid-size | full", "f64", 512, 48, MID), ("mid-size | full", "f64", 512, 49, MID),
    ("diagonal tile inside the panel launches | launches of its own", "f64", 2048, 256, FUS),
    ("diagonal tile inside the panel launches | launches of its own", "f64", 2048, 512, FUS),
]
QUICK = [0, 1, 6, 7, 10, 11, 12, 13]


def child(dt, N, B):
    import torch
    import bench
    import corenav_gp_amd.engine as engine
    import corenav_gp_amd.synth as synth
    dev = torch.device("cuda", 0)
    kid, X, y, Xs, th, dts = synth.config(2 if dt == "f64" else 3, batch=B, N=N, M=bench.M_TEST)
    w = bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, dts, 1)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        w.step()
        torch.cuda.synchronize()
    reps = 30 if B < 200 else 8
    best = 1e9
    for _ in range(3):       # best of three back-to-back bursts
        t0 = time.perf_counter()
        for _ in range(reps):
            w.step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    assert int(w.dinfo.abs().sum().item()) == 0
    print(json.dumps({"ms": best}))


def measure(dt, N, B, env):
    e = dict(os.environ, CGP_LIB=AB)
    for k in ("CGP_SCHED", "CGP_LAT_FITS", "CGP_MID_FITS"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", dt, str(N), str(B)], env=e, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"{dt} N={N} B={B} {env}: {r.stderr[-800:]}")
    return json.loads(lines[-1])["ms"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tol", type=float, default=0.07)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--child", nargs=3)
    a = ap.parse_args()
    if a.child:
        child(a.child[0], int(a.child[1]), int(a.child[2]))
        return 0
    if not os.path.exists(AB):
        print("libcorenav_gp_ab.so not built (make -C corenav_gp_amd/csrc ab)")
        return 2
    pts = [POINTS[i] for i in QUICK] if a.quick else POINTS
    bad = 0
    print(f"# shipped schedule choice against the switch forced either way, ms per call (same box, fresh process each); tolerance {a.tol:.0%}")
    for sw, dt, N, B, alts in pts:
        shipped = measure(dt, N, B, {})
        alt = {name: measure(dt, N, B, env) for name, env in alts.items()}
        best = min(alt.values())
        ok = shipped <= (1.0 + a.tol) * best
        bad += 0 if ok else 1
        print(f"{dt} N={N:5d} fits={B:4d}  shipped {shipped:8.3f}   " + "   ".join(f"{k} {v:8.3f}" for k, v in alt.items()) +
              f"   shipped / best {shipped / best:5.3f}  {'ok' if ok else 'SLOWER THAN THE ALTERNATIVE'}   [{sw}]", flush=True)
    print(f"# {len(pts) - bad} of {len(pts)} points within tolerance")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
