#!/usr/bin/env python3
"""Condenses a rocprofv3 kernel_trace.csv into the timeline of the last call: one line per launch (short kernel name, grid, queue,
start and end in us relative to the call's first launch, duration), calls being delimited by k_prep launches."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def col(r, *names):
    for n in names:
        if n in r: return r[n]
    return ""
recs = []
for r in rows:
    name = col(r, "Kernel_Name")
    s, e = int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))
    grid = "x".join(str(int(col(r, f"Grid_Size_{a}") or 1) // max(1, int(col(r, f"Workgroup_Size_{a}") or 1))) for a in "XYZ")
    recs.append((s, e, name, grid, col(r, "Queue_Id"), col(r, "Stream_Id")))
recs.sort()
starts = [i for i, r in enumerate(recs) if "k_prep" in r[2]]
if len(starts) < 3:
    print("no k_prep launches found"); sys.exit(1)
a, b = starts[-2], starts[-1]
call = recs[a:b]
t0 = call[0][0]
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
print(f"{'kernel':44s} {'grid':>12s} {'queue':>6s} {'start':>8s} {'end':>8s} {'dur':>7s}")
for s, e, n, g, q, st in call:
    print(f"{short(n):44s} {g:>12s} {q:>6s} {(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}")
print(f"call span {(max(r[1] for r in call) - t0) / 1e3:.1f} us; period to the next call's k_prep {(recs[b][0] - t0) / 1e3:.1f} us")
