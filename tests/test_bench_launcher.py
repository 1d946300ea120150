"""CPU test of bench.py's own launcher: `python bench.py --gpus 2` (no torchrun) must start two fresh rank
processes, rendezvous (gloo here, RCCL on the GPU box), take the max over ranks and print ONE line whose
n_gpus is 2.  CGP_BENCH_DRY=1 replaces the GPU step by a sleep: the line is marked dry_run and is never a
measurement -- this covers the launch / barrier / gather plumbing only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, **env):
    e = dict(os.environ, CGP_BENCH_BACKEND="gloo", CGP_BENCH_DRY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=300, cwd=ROOT, env=e)


def test_gpus2_spawns_two_ranks():
    r = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["dry_run"] is True
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and j["config"]["collective_backend"] == "gloo"
    assert len(j["config"]["per_rank_fits_per_s"]) == 2 and all(v > 0 for v in j["config"]["per_rank_fits_per_s"])
    assert j["config"]["ensemble"]["n"] == 16           # both ranks' summaries were gathered
    assert j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    # whole-job value = fits of all ranks / max-over-ranks time
    assert j["value"] <= sum(j["config"]["per_rank_fits_per_s"]) * 1.001


def test_world_size_must_match_gpus():
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_rank_dry_line():
    r = run_bench("--steps", "2", "--warmup", "0", "--batch", "4")
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["ranks"] == 1 and j["config"]["collective_backend"] is None
