#!/bin/bash
# SQ / TCC counters of one bench configuration (separate --pmc passes, as the MI355X guide prescribes):
#   tools/pmc_cfg.sh <tag> <bench args...>   -> gpurun_out/<tag>_pmcN/
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${tag}_pmc$i -o p -- python3 $R/bench.py --no-pmc --steps 1 --warmup 1 --no-cpu --no-extra "$@" > $O/${tag}_pmc$i.log 2>&1
done
ls $O/${tag}_pmc*/ | head
