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
    # a plain multi-rank run also carries BASELINE configs[2] as written (strong scaling, one context per GPU)
    c3 = j["config"]["extra"]["cfg3_strong"]
    assert c3["n_gpus"] == 2 and c3["scaling"] == "strong" and c3["pipeline_depth"] == 1 and c3["fits_per_step_all_ranks"] == 512
    assert c3["fits_per_gpu_per_call"] == 256 and c3["fits_per_s"] > 0


def test_world_size_must_match_gpus():
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_rank_dry_line():
    r = run_bench("--steps", "2", "--warmup", "0", "--batch", "4")
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["ranks"] == 1 and j["config"]["collective_backend"] is None


def test_strong_scaling_shards_the_total():
    """--scaling strong: --batch is the sweep's TOTAL (BASELINE configs[2]: 512 fits over the ranks); two ranks own
    contiguous shards of 5 and 4 of 9 fits, the gathered table has 9 rows, value counts 9 fits per step."""
    r = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "9", "--scaling", "strong", "--config", "3")
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["ensemble"]["n"] == 9
    assert j["config"]["pipeline_depth"] == 1      # the single-context figure is the default; --pipeline 2 is cross-sweep overlap
    assert j["config"]["fits_per_step_all_ranks"] == 9 and j["config"]["fits_per_gpu_per_step"] == 4.5
    # per-rank rates are in the ratio of the shard sizes (5 : 4) up to timing noise, and sum to about the whole-job value
    a, b = j["config"]["per_rank_fits_per_s"]
    assert 0.8 < a / b < 2.0      # 5 : 4 up to timing noise of a 2 ms sleep-step on a busy host
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 9) < 1e-6


def test_forced_one_rank_world_runs_the_collectives():
    """CGP_BENCH_FORCE_DIST=1: a world of ONE rank still initialises the process group and runs barrier / all_gather /
    all_reduce / the summary gather (gloo here; RCCL in tests/test_gpu_bench_contract.py)."""
    r = run_bench("--steps", "2", "--warmup", "0", "--batch", "4", CGP_BENCH_FORCE_DIST="1")
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["collective_backend"] == "gloo" and j["config"]["collective_world"] == 1
    assert j["config"]["ensemble"]["n"] == 4


def test_a_rank_that_dies_ends_the_job_quickly():
    """One rank failing at start-up must not leave the others waiting in the rendezvous: the launcher kills them and
    returns the failing code."""
    import time
    t0 = time.time()
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4", CGP_BENCH_TEST_DIE_RANK="1")
    assert r.returncode != 0
    assert time.time() - t0 < 60


def test_live_pmc_passes_are_skipped_under_a_profiler(monkeypatch):
    """bench.py started under rocprofv3 / rocprofiler-sdk (tool library preloaded, ROCP_* / ROCPROF* set) must not start
    profilers of its own: the nested start-up would exec out of a GPU-initialised process and pollute the outer counters."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for k in list(os.environ):
        if k == "LD_PRELOAD" or k.startswith(bench.PROFILER_ENV_PREFIXES):
            monkeypatch.delenv(k, raising=False)
    assert bench.under_profiler() is False
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler() is True
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler() is True
    called = []
    monkeypatch.setattr(bench.subprocess, "Popen", lambda *a, **k: called.append(a) or (_ for _ in ()).throw(AssertionError("spawned")))
    assert bench.pmc_traffic_live(bench.parse_args([])) is None and not called


def test_tools_never_profile_bench_with_live_pmc():
    """Every tools/*.sh that wraps bench.py in rocprofv3 passes --no-pmc."""
    import glob
    import re
    for f in glob.glob(os.path.join(ROOT, "tools", "*.sh")):
        for line in open(f):
            if "rocprofv3" in line and "bench.py" in line:
                assert re.search(r"bench\.py\s+--no-pmc", line), (f, line)
