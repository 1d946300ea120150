#!/bin/bash
# Diagnostics of one bench configuration on the GPU box (run through gpurun from the repo root):
#   tools/diag_cfg.sh <tag> <bench args...>
# writes under gpurun_out/<tag>_*: the plain bench line, the per-launch HIP-event dump (CGP_PROF_DUMP),
# timing ablations (CGP_DBG bits 64 = no trmm, 128 = no exp, 192 = both; results wrong on purpose; only
# in a -DCGP_ABLATION build) and the rocprofv3 kernel-trace summary.
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-pmc --no-cpu --no-extra "$@" > $O/${tag}_bench.json 2> $O/${tag}_bench.err
CGP_PROF_DUMP=1 python3 $R/bench.py --no-pmc --no-cpu --no-extra --steps 2 --warmup 1 "$@" > $O/${tag}_dump.json 2> $O/${tag}_dump.err
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
for dbg in 64 128 192; do
  CGP_LIB=$AB CGP_DBG=$dbg python3 $R/bench.py --no-pmc --no-cpu --no-extra "$@" > $O/${tag}_dbg$dbg.json 2> $O/${tag}_dbg$dbg.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o $tag -- python3 $R/bench.py --no-pmc --no-cpu --no-extra --steps 5 --warmup 2 "$@" > $O/${tag}_stats.log 2>&1
for f in $O/${tag}_bench.json $O/${tag}_dbg64.json $O/${tag}_dbg128.json $O/${tag}_dbg192.json; do tail -1 $f | cut -c1-200; done
