#!/bin/bash
# Per-launch HIP-event times (CGP_PROF_DUMP) of the working-tree and the prev library:
#   tools/dump_launches.sh <tag> <bench args...>  -> gpurun_out/<tag>_dump_{new,prev}.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in new prev; do
  if [ $v = new ]; then lib=$R/corenav_gp_amd/libcorenav_gp.so; else lib=$R/corenav_gp_amd/libcorenav_gp_prev.so; fi
  CGP_LIB=$lib CGP_PROF_DUMP=1 python3 $R/bench.py --no-pmc --no-cpu --no-extra --steps 2 --warmup 1 "$@" 2>&1 >/dev/null | grep "cgp prof" > $R/gpurun_out/${tag}_dump_$v.txt
done
