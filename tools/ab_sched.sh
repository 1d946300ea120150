#!/bin/bash
# A/B of the alternative schedules (-DCGP_AB library) on one box: tools/ab_sched.sh <tag> <bench args...>
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
run() { name=$1; shift; env "$@" CGP_LIB=$AB python3 $R/bench.py --no-pmc --no-cpu --no-extra "${ARGS[@]}" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', round(j['value'],1), round(j['ms_per_step'],3), j['kernel_ms_per_step'])"; }
ARGS=("$@")
for rep in 1 2; do
  run default CGP_X=0
  run overlap CGP_SCHED=overlap
  run fuseddiag CGP_SCHED=fuseddiag
  ARGS=("$@" --streams 2); run streams2 CGP_X=0
  ARGS=("$@" --streams 4); run streams4 CGP_X=0
  ARGS=("$@")
done 2>&1 | tee $O/${tag}_absched.txt
