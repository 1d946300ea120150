"""CPU test of the multi-device sweep's host side (csrc/cgp_sweep.cpp: persistent per-device worker threads, shard 0 on the
caller's thread, both entry points) under ThreadSanitizer: the file is compiled against a test double of the engine entry
points it calls (tests/sweep_doubles/engine_double.cpp -- no GPU, no HIP) and driven through 1 200 calls over five
"devices", batches from 1 to 64 (empty shards included), creation and destruction three times.  Any data race, a shard run
twice or on the wrong context, a call that returns before a slow shard finished, or a lost per-fit status fails it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_sweep_worker_threads_under_sanitizer(tmp_path, sanitizer):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / f"sweep_{sanitizer}")
    src = [os.path.join(ROOT, "corenav_gp_amd", "csrc", "cgp_sweep.cpp"), os.path.join(ROOT, "tests", "sweep_doubles", "engine_double.cpp"),
           os.path.join(ROOT, "tests", "sweep_doubles", "driver.cpp")]
    r = subprocess.run([cxx, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-o", exe, *src, "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip(f"-fsanitize={sanitizer} not usable here: {r.stderr[-200:]}")
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "sweep threads ok" in run.stdout, (run.returncode, run.stdout[-500:], run.stderr[-3000:])
