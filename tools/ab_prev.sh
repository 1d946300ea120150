#!/bin/bash
# Same-box A/B of the working-tree library against libcorenav_gp_prev.so (tools/mk_prev.sh):
#   tools/ab_prev.sh <tag> <bench args...>     alternates new / prev three times
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  for v in new prev; do
    if [ $v = new ]; then lib=$R/corenav_gp_amd/libcorenav_gp.so; else lib=$R/corenav_gp_amd/libcorenav_gp_prev.so; fi
    echo -n "$v: "; CGP_LIB=$lib timeout 300 python3 $R/bench.py --no-pmc --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['ms_per_step'],3), {k: round(v,3) for k,v in j['kernel_ms_per_step'].items()}, round(j['roofline']['frac'],4), round(j['config']['single_fit_latency_ms'],3))"
  done
done 2>&1 | tee $R/gpurun_out/${tag}_abprev.txt
