#!/bin/bash
# Same-box A/B of the sliding-window kernel between library builds: tools/ab_window.sh <tag> lib1.so lib2.so ...
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "; CGP_LIB=$R/corenav_gp_amd/$lib python3 $R/tools/bench_window.py --windows 1024 --ticks 200 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['frac'],4), round(j['us_per_tick_per_window'],1))"
  done
done 2>&1 | tee $R/gpurun_out/${tag}_abwin.txt
