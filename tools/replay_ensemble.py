#!/usr/bin/env python3
"""Monte-Carlo closed-loop replay sharded over the GPUs of a node (stand-in for BASELINE configs[4]):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/replay_ensemble.py --traj 64
Each rank replays its block of trajectories (corenav_gp_amd/replay.py); the only collective is the
all-gather of per-trajectory summaries (first stop time, number of windows / stops)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--traj", type=int, default=32)
ap.add_argument("--ticks", type=int, default=900)
args = ap.parse_args()
import torch
import torch.distributed as dist
from corenav_gp_amd import replay, sharding, synth

rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
# CGP_BENCH_SAME_DEVICE=1 + CGP_BENCH_BACKEND=gloo: every rank on GPU 0, summaries over gloo -- the two-rank
# self-test of a one-GPU box (RCCL refuses two ranks on one device); as bench.py
backend = os.environ.get("CGP_BENCH_BACKEND", "nccl")
if os.environ.get("CGP_BENCH_SAME_DEVICE"):
    local = 0
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
cdev = dev if backend == "nccl" else torch.device("cpu")
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
a, b = sharding.shard_range(args.traj, rank, world)
ens = replay.ClosedLoopEnsemble(n_traj=b - a, device=local, seed=synth.SEED_BASE + 5 + 31 * a)
t0 = time.perf_counter()
npub = ens.run(args.ticks)
dt = time.perf_counter() - t0
local_tab = torch.tensor([[tr.stop_cmds[0] if tr.stop_cmds else -1.0, float(len(tr.windows)), float(tr.stops)]
                          for tr in ens.traj], dtype=torch.float64, device=cdev)
table = sharding.gather_summaries(local_tab, args.traj)
if rank == 0:
    tab = table.cpu().numpy()
    print(json.dumps({"trajectories": args.traj, "ticks": args.ticks, "n_gpus": world, "wall_s": dt,
                      "per_trajectory": tab.tolist(),
                      "windows": int(tab[:, 1].sum()), "stops": int(tab[:, 2].sum()),
                      "first_stop_cmd_mean_s": float(tab[tab[:, 0] >= 0, 0].mean()) if (tab[:, 0] >= 0).any() else None}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
