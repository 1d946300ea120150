import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench, corenav_gp_amd.engine as engine, corenav_gp_amd.synth as synth
for B in (1, 4, 64, 512):
    kid, X, y, Xs, th, dts = synth.config(3, batch=B)
    X = X[:, :, :1].copy(); Xs = Xs[:, :, :1].copy(); th = np.ascontiguousarray(th[:, [0, 1, th.shape[1] - 1]])
    W = bench.Workload(engine, torch, torch.device("cuda", 0), 0, 0, X, y, Xs, th, dts, 0)
    for _ in range(12): W.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): W.step()
    torch.cuda.synchronize(); print("d=1 fp32 N=1024 batch", B, round((time.perf_counter() - t0) / 20 * 1e3, 4), "ms per call")
