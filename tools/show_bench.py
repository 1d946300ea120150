#!/usr/bin/env python3
"""Prints the headline and the numeric extras of a bench.py JSON line: python tools/show_bench.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("roofline", {}).get("frac"))
x = d["config"].get("extra", {})
for k, v in x.items():
    if isinstance(v, (int, float)):
        print(" ", k, v)
for k in ("cgp_sweep", "replay", "cgp_sweep_error", "replay_error"):
    if k in x:
        print(" ", k, x[k])
