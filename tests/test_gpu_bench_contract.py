"""bench.py prints ONE JSON line with the driver's contract fields (plus roofline and cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "1", "--no-extra", "--no-pmc"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"] == "GP-fits/s" and j["unit"] == "fits/s" and j["n_gpus"] == 1 and j["steps"] == 2
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf and "from" in rf
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["gpu_vs_oracle_max_rel_err"] < 1e-6
    assert cb["host"]["cpu_model"] and "governor" in cb["host"] and cb["port_threads_over_batch"]["value"] > 0
    assert "cgroup_cpu_max" in cb["host"] and "lapack_all_threads" not in cb
    rows = cb["single_thread_rows"]      # the baseline is the FASTER of the two single-thread runs of the oracle
    assert cb["value"] == max(r["value"] for r in rows.values() if "value" in r)
    assert "ablation_build" not in j and "dry_run" not in j
    assert j["value"] > 0 and j["config"]["single_fit_latency_ms"] > 0


def test_bench_extra_lines():
    """config.extra of the same driver command: the PCIe-inclusive rate, BASELINE configs[2] (512 x N=1024
    fp32) on its own schedule, configs[3] (sliding window) and the batched look-ahead."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "1", "--warmup", "1",
                        "--cpu-sample", "1", "--no-pmc"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    ex = j["config"]["extra"]
    assert not [k for k in ex if k.endswith("_error")], ex
    assert ex["end_to_end_fits_per_s"] > 0 and ex["end_to_end_matches_resident"] is True
    assert ex["cfg3_fits_per_s"] > 0 and 0 < ex["cfg3_roofline_frac"] < 1 and ex["cfg3_max_rel_err_vs_oracle"] < 1e-3
    assert ex["cfg3_cpu_baseline"]["value"] > 0 and ex["cfg3_cpu_baseline"]["kind"] == "port" and "_cpu_leg" not in ex
    assert ex["window_ticks_per_s"] > 0 and 0 < ex["window_hbm_frac"] < 1 and ex["lookahead_traj_per_s"] > 0
    assert ex["window_ticks_per_pass"] == 4 and ex["window_hbm_frac_one_pass_per_tick_equivalent"] == pytest.approx(4 * ex["window_hbm_frac"])
    assert 0 < ex["window_host_tick_us"] < 2000
    assert 0 < ex["node_callback_us"] < 5000 and 0 < ex["node_callback_opt_ms"] < 500   # the reference node's own work item
    # configs[2] as written (64 fits per GPU at 8 GPUs): the 64-fit call rate and the projection from this GPU's two rates
    assert ex["cfg3_fits_per_s_at_64"] > 0 and 0 < ex["cfg3_strong_scaling_projection_8gpu"] <= 8.5
    # the C-ABI multi-device entry on the same shard: over one device it IS the context's call (shard 0 runs on the caller's thread)
    sw = ex["cgp_sweep"]
    assert 0.9 < sw["devices_1_over_context"] < 1.15 and sw["devices_same_gpu_twice_ms_per_call"] > 0
    assert ex["replay"]["windows_fitted"] >= 64 and ex["replay"]["stops"] >= 1
    assert ex["cfg3_strong"]["n_gpus"] == 1 and ex["cfg3_strong"]["fits_per_s"] == ex["cfg3_fits_per_s"]
    # round 6: what cfg3_roofline_frac is (a label against the FP32-input MFMA peak) next to the pipe the loop runs on and HBM, over 20 steps
    assert ex["cfg3_steps"] == 20 and 0 < ex["cfg3_frac_of_bf16x6_bound"] < ex["cfg3_roofline_frac"] and 0 < ex["cfg3_hbm_frac"] < 1
    assert "window_hbm_frac_counter_traffic_projected" in ex and "window_hbm_frac_counter_traffic" not in ex
    # BASELINE configs[1] read literally -- ONE fit -- priced on the line
    assert 0 < j["config"]["single_fit_cholesky_roofline_frac"] < 0.2 and j["config"]["single_fit_vs_cpu_1thread"] > 10
    # fp32: the mean refined against a double-precision residual (the CPU leg's oracle check)
    fr = ex["fp32_refinement"]
    assert fr["mean_err_vs_oracle_default_max"] < 2e-5 < fr["mean_err_vs_oracle_unrefined_max"] and fr["variance_err_vs_oracle_max"] < 1e-3


def test_bench_live_pmc_traffic():
    """roofline.traffic is measured by the run itself: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) of
    the same command.  32 fits per call take the throughput schedule, so k_panel launches exist."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "32", "--steps", "1", "--warmup", "1",
                        "--no-cpu", "--no-extra"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rf = json.loads(lines[0])["roofline"]
    assert rf["traffic_source"].startswith("rocprofv3 --pmc"), rf
    # every tile streams its panels at least once: more than the 26 MB per fit of L written once, less than 10x the streaming figure
    assert 32 * 26e6 / 16 < rf["traffic"] < 32 * 3e9 / 16, rf


def test_bench_gpus2_on_one_gpu():
    """`bench.py --gpus 2` starts two ranks by itself; on a one-GPU box both use device 0
    (CGP_BENCH_SAME_DEVICE) and the summaries travel over gloo (RCCL refuses two ranks on one GPU)."""
    env = dict(os.environ, CGP_BENCH_SAME_DEVICE="1", CGP_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and len(j["config"]["per_rank_fits_per_s"]) == 2
    assert j["config"]["ensemble"]["n"] == 32 and j["config"]["ensemble"]["n_failed"] == 0
    assert "cpu_baseline" not in j and j["roofline"]["frac"] > 0
    c3 = j["config"]["extra"]["cfg3_strong"]   # BASELINE configs[2] as written over the same ranks (256 fits per rank and call here)
    assert c3["n_gpus"] == 2 and c3["fits_per_gpu_per_call"] == 256 and c3["pipeline_depth"] == 1 and c3["fits_per_s"] > 0


def test_bench_strong_scaling_two_ranks_on_one_gpu():
    """BASELINE configs[2] as written, two ranks: `--scaling strong --config 3 --batch 64` cuts the 64 fits into two
    contiguous shards of 32 (one engine context each, both on device 0 here), the gathered table has 64 rows and
    `value` counts 64 fits per step."""
    env = dict(os.environ, CGP_BENCH_SAME_DEVICE="1", CGP_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--config", "3",
                        "--batch", "64", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["dtype"] == "f32"
    assert j["config"]["fits_per_step_all_ranks"] == 64 and j["config"]["fits_per_gpu_per_step"] == 32
    assert j["config"]["ensemble"]["n"] == 64 and j["config"]["ensemble"]["n_failed"] == 0
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 64) < 1e-6


def test_bench_rccl_collectives_with_one_rank():
    """CGP_BENCH_FORCE_DIST=1: `bench.py --gpus 1` initialises the `nccl` backend (= RCCL on ROCm) with a world of one
    rank and runs the barrier, the all_gather / all_reduce of the timings and sharding.gather_summaries ON THE DEVICE:
    librccl loads, HSA_ENABLE_IPC_MODE_LEGACY handling and the gather are exercised on an MI355X before an 8-GPU job
    is the first to try them."""
    env = dict(os.environ, CGP_BENCH_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "CGP_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "3", "--batch", "64", "--steps", "3",
                        "--warmup", "1", "--no-cpu", "--no-extra", "--no-pmc"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["config"]["collective_backend"] == "rccl" and j["config"]["collective_world"] == 1
    assert j["n_gpus"] == 1 and j["config"]["ensemble"]["n"] == 64 and j["config"]["ensemble"]["n_failed"] == 0
    # an fp32 line of the shipped library says on which matrix cores its products run, and prices the kernel against both ceilings
    rf = j["roofline"]
    assert j["dtype"] == "f32" and rf["peak"] == pytest.approx(157.3) and "bf16x6" in rf["mfma_path"]
    assert rf["bf16x6_bound_tflops"] == pytest.approx(2500.0 / 6.0) and rf["frac_of_bf16x6_bound"] == pytest.approx(rf["achieved"] / (2500.0 / 6.0))
