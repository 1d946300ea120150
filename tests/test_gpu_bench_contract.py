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
                        "--cpu-sample", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
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
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["gpu_vs_oracle_max_rel_err"] < 1e-6
    assert j["value"] > 0 and j["config"]["single_fit_latency_ms"] > 0
