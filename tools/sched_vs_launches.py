#!/usr/bin/env python3
"""One-launch schedule (k_sched, -DCGP_AB library, CGP_SCHED=onelaunch) against the launches the engine ships: the same
mid-size call through both, in two child processes (CGP_SCHED is read once per process); results must be bitwise equal.
usage: sched_vs_launches.py [f32|f64] N batch      (prints one line: ms per call of both, fits that differ)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch, time
    import bench
    import corenav_gp_amd.engine as engine, corenav_gp_amd.synth as synth
    dt, N, B, reps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dev = torch.device("cuda", 0)
    kid, X, y, Xs, th, dts = synth.config(2 if dt == "f64" else 3, batch=B, N=N, M=bench.M_TEST)
    w = bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, dts, 1)
    for _ in range(int(os.environ.get("SVL_WARM", "3"))):
        w.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        w.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    np.savez(sys.argv[6], mean=w.dmean.cpu().numpy(), var=w.dvar.cpu().numpy(), logml=w.dlogml.cpu().numpy(), info=w.dinfo.cpu().numpy(), ms=ms)
    sys.exit(0)
import numpy as np
dt, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
res = {}
for mode in ("sched", "launches"):
    out = f"/tmp/svl_{mode}.npz"
    env = dict(os.environ, CGP_LIB=os.path.join(ROOT, "corenav_gp_amd", "libcorenav_gp_ab.so"))
    env.pop("CGP_SCHED", None)
    if mode == "sched":
        env["CGP_SCHED"] = "onelaunch"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", dt, str(N), str(B), os.environ.get("SVL_REPS", "20"), out], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(mode, "FAILED", r.stderr[-1500:])
        sys.exit(1)
    res[mode] = np.load(out)
a, b = res["sched"], res["launches"]
badl = [i for i in range(B) if a["logml"][i] != b["logml"][i]]
badm = [i for i in range(B) if not np.array_equal(a["mean"][i], b["mean"][i])]
print("  logml differs:", len(badl), badl[:8], " mean differs:", len(badm), badm[:8])
bad = [i for i in range(B) if not (np.array_equal(a["mean"][i], b["mean"][i]) and np.array_equal(a["var"][i], b["var"][i]) and a["logml"][i] == b["logml"][i])]
print(f"{dt} N={N} B={B}: sched {float(a['ms']):.3f} ms  launches {float(b['ms']):.3f} ms  info(sched) nonzero: {int(np.count_nonzero(a['info']))}  "
      f"fits that differ: {len(bad)} {bad[:10]}  max |dlogml| {float(np.max(np.abs(a['logml'] - b['logml']))):.3e}")
