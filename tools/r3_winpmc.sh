#!/bin/bash
# HBM traffic of the sliding-window bench (1024 windows, 200 ticks): separate FETCH_SIZE / WRITE_SIZE passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r3w_pmc$i -o p -- python3 $R/tools/bench_window.py --windows 1024 --ticks 200 > $O/r3w_pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/r3w_pmc*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_window" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("cgp::")[1].split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in tot:
    print(k, {c: (round(v * 1024 / 1e9, 2), n[k][c]) for c, v in tot[k].items()}, "(GB summed over dispatches, dispatch count; FETCH x2 for gfx950 not applied)")
PY
