#!/bin/bash
# Round-6 evidence run (GPU box, through gpurun from the repo root): rocprofv3 kernel stats + PMC passes of the default bench
# command and of configs[2], the sliding-window PMC passes, the 64-fit timeline with default stream groups, a refined d = 1 call's
# timeline and the refinement's cost table.
#   bash tools/profile_r06.sh   -> gpurun_out/r6f_* (condensed into profiles/r06_* by tools/summarize_profiles.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r6f > gpurun_out/r6f_profile.log 2>&1
bash tools/profile_round.sh r6f32 --config 3 > gpurun_out/r6f32_profile.log 2>&1
bash tools/pmc_window.sh r6w > gpurun_out/r6w_profile.log 2>&1
bash tools/mid_timeline.sh r6f_mid > gpurun_out/r6f_mid.log 2>&1
bash tools/mid_timeline.sh r6f_mid1 --streams 1 > gpurun_out/r6f_mid1.log 2>&1
python3 tools/refine_cost.py > gpurun_out/r6f_refine_cost.txt 2>&1
python3 tests/fuzz/d1_fp32_error.py > gpurun_out/r6f_d1_fp32_error.txt 2>&1
python3 tests/fuzz/d1_fp32_error.py --refine 0 > gpurun_out/r6f_d1_fp32_error_unrefined.txt 2>&1
python3 tests/fuzz/rho_vs_error.py > gpurun_out/r6f_rho_vs_error.txt 2>&1
python3 tools/summarize_profiles.py r6f r06 > /dev/null 2>&1
python3 tools/summarize_profiles.py r6f32 r06 512 1024 f32 > /dev/null 2>&1
cp profiles/r06_kernel_stats.csv profiles/r06_kernel_stats_f32.csv profiles/r06_pmc_summary.json profiles/r06_pmc_summary_f32.json gpurun_out/ 2>/dev/null
tail -1 gpurun_out/r6f_bench.json | cut -c1-300
