"""A short prefix of the randomised parity sweep (tests/fuzz/fuzz_parity.py: window lengths around every tile and
schedule boundary, horizons around multiples of 128, all three kernels, both precisions, call sizes either side of
the latency / throughput and fused / split switches) against the oracle.  The seeded case sequence is fixed; the
time budget only decides how long a prefix of it runs (the builder ran 150 s of this seed clean)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_prefix():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_parity.py"), "20", "7"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("cases ") and " failures 0 " in last, last
    assert int(last.split()[1]) >= 20, last


def test_fuzz_single_window_api_prefix():
    """tests/fuzz/fuzz_single.py: cgp_fit -> cgp_predict (with / without noise) -> cgp_get_alpha -> cgp_get_factor ->
    cgp_nll_grad on random windows, fp64, against the oracle (the builder ran 120 s of this seed: 2 036 cases clean)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_single.py"), "12", "1"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("cases ") and " failures 0 " in last, last


def test_fuzz_sliding_window_prefix():
    """tests/fuzz/fuzz_window.py: random window lengths around the 16-column panel boundaries, kernels, dimensions, stream
    lengths with several ring compactions, 1-3 windows, random block cuts, against the refit-per-tick oracle (the
    builder ran 90 s of this seed: 2 283 cases clean)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_window.py"), "10", "1"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("cases ") and " failures 0 " in last, last


def test_fuzz_lookahead_prefix():
    """tests/fuzz/fuzz_lookahead.py: the batched GPU look-ahead against the host C++ path of the same ABI on random horizons
    (0 ... 748 predictions), slip levels, filter snapshots, thresholds, late arrivals, both H packings (the builder
    ran 60 s of this seed: 1 139 ensembles clean)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_lookahead.py"), "8", "1"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert " failures 0" in r.stdout.strip().splitlines()[-1], r.stdout[-500:]


def test_fuzz_short_window_kernel_prefix():
    """tests/fuzz/fuzz_small.py: the one-launch short-window kernel on every window length 2 ... 160, d = 1 ... 8, the three
    kernels -- value, gradient and GPy's jitter against the oracle, cgp_predict after it (lazy refit), and the device
    L-BFGS through cgp_optimize / cgp_optimize_batch (the builder ran 150 s of this seed clean)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_small.py"), "12", "1"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("cases ") and " failures 0 " in last, last
