"""A mid-size fp64 call puts its extra-row launches on a worker stream that must run BESIDE the caller's stream.  Until
round 4 that depended on which hardware queue the worker stream happened to share (a 64-fit call right after a 512-fit
context had been created and stepped: 8.34 instead of 6.87 ms, profiles/r04_ctx_placement.txt).  The worker streams now
live at the lowest stream priority (cgp_create): the call must cost the same whatever was created before it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mid_size_call_does_not_depend_on_what_was_created_before():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ctx_placement.py"), "--dtype", "f64", "fresh", "after512",
                        "after512_s1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {j["scenario"]: j for j in (json.loads(l) for l in r.stdout.splitlines() if l.startswith("{"))}
    assert not [j for j in rows.values() if "error" in j], rows
    fresh = rows["fresh"]["ms_per_64_fit_call"]
    for scn in ("after512", "after512_s1"):
        assert rows[scn]["ms_per_64_fit_call"] <= 1.05 * fresh, (scn, rows[scn]["ms_per_64_fit_call"], fresh)
