#!/bin/bash
# Profiles the default bench command on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench args...]
# writes gpurun_out/<tag>_stats (rocprofv3 --kernel-trace --stats), gpurun_out/<tag>_pmc1..4 (one
# counter group per run, as the MI355X guide prescribes; never combined with sys/hip/hsa tracing) and
# gpurun_out/<tag>_bench.json (the plain bench line).  Condense afterwards with:
#   python tools/summarize_profiles.py <tag> rNN [batch] [N] [dtype]
set -u
tag=${1:-r2}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o $tag -- python3 $R/bench.py --no-pmc --no-cpu --no-extra "$@" > $R/gpurun_out/${tag}_stats.log 2>&1
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc$i -o p -- python3 $R/bench.py --no-pmc --steps 1 --warmup 1 --no-cpu --no-extra "$@" > $R/gpurun_out/${tag}_pmc$i.log 2>&1
done
tail -1 $R/gpurun_out/${tag}_bench.json | cut -c1-300
