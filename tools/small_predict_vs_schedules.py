#!/usr/bin/env python3
"""Short windows (N <= 160, fp64) through cgp_fit_predict_batch_device: the one-launch LDS kernel (k_small_predict) against the
tiled schedules (CGP_SMALLPRED=off in the measurement library), per batch size.  Fresh process per point.
usage: python3 tools/small_predict_vs_schedules.py        (needs libcorenav_gp_ab.so)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "corenav_gp_amd", "libcorenav_gp_ab.so")


def child(cfg, N, B):
    import torch, bench
    import corenav_gp_amd.engine as engine
    import corenav_gp_amd.synth as synth
    dev = torch.device("cuda", 0)
    kid, X, y, Xs, th, dts = synth.config(cfg, batch=B, N=N, M=bench.M_TEST)
    w = bench.Workload(engine, torch, dev, 0, kid, X, y, Xs, th, "f64", 1)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        w.step(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20): w.step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print(json.dumps({"ms": best, "mean0": float(w.dmean[0, 0].item()), "var_last": float(w.dvar[B - 1, -1].item()), "logml": float(w.dlogml[B - 1].item()),
                      "bad": int(w.dinfo.abs().sum().item())}))


def run(cfg, N, B, env):
    e = dict(os.environ, CGP_LIB=AB); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(cfg), str(N), str(B)], env=e, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not lines: raise RuntimeError(r.stderr[-600:])
    return json.loads(lines[-1])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])); sys.exit(0)
    print("# ms per call, 599 predictions per window, fp64; cfg 1 = SE-iso d = 3, cfg 2 = SE-ARD d = 6")
    for cfg, N in ((1, 134), (1, 64), (2, 134), (2, 160)):
        for B in (1, 8, 64, 256, 512):
            a, b = run(cfg, N, B, {}), run(cfg, N, B, {"CGP_SMALLPRED": "off"})
            same = abs(a["mean0"] - b["mean0"]) < 1e-6 * (1 + abs(b["mean0"])) and abs(a["logml"] - b["logml"]) < 1e-6 * (1 + abs(b["logml"])) and abs(a["var_last"] - b["var_last"]) < 1e-6 * (1 + abs(b["var_last"]))
            print(f"cfg{cfg} N={N:4d} windows={B:4d}   one launch {a['ms']:8.3f}   schedules {b['ms']:8.3f}   ratio {b['ms'] / a['ms']:5.2f}x   "
                  f"{B / a['ms']:9.1f} k windows/s   results agree: {same}  (info {a['bad']}/{b['bad']})", flush=True)
