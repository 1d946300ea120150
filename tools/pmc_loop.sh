#!/bin/bash
# Issue / wait / LDS counters of the panel kernel for one library build (GPU box, through gpurun from the repo root):
#   tools/pmc_loop.sh <tag> <lib name under corenav_gp_amd/, e.g. gp_bx6> [bench args...]
# one counter group per rocprofv3 run (never combined with sys/hip/hsa tracing); prints per-kernel sums for k_panel.
tag=$1; lib=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export CGP_LIB=$R/corenav_gp_amd/libcorenav_$lib.so
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc$i -o p -- python3 $R/bench.py --no-pmc --steps 1 --warmup 1 --no-cpu --no-extra "$@" > $R/gpurun_out/${tag}_pmc$i.log 2>&1
done
python3 - <<P
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/${tag}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_panel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(f"{k:36s} {tot[k]:.4g}")
P
