#!/bin/bash
# rocprofv3 kernel stats + PMC passes of a refined fp32 call (512 x N = 1024, d = 1, M = 599): tools/profile_refine.sh <tag>
tag=${1:-r6ref}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o $tag -- python3 $R/tools/refine_call.py > $O/${tag}_stats.log 2>&1
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${tag}_pmc$i -o p -- python3 $R/tools/refine_call.py --reps 2 > $O/${tag}_pmc$i.log 2>&1
done
cp $O/${tag}_stats/${tag}_kernel_stats.csv $O/${tag}_kernel_stats.csv
