#!/bin/bash
# Round-5 evidence run (GPU box, through gpurun from the repo root): rocprofv3 kernel stats + PMC passes of the default bench
# command and of configs[2], the sliding-window PMC passes, and the stream-group probes.
#   bash tools/profile_r05.sh   -> gpurun_out/r5f_* (condense with tools/summarize_profiles.py r5f r05 / r5f32 r05 512 1024 f32)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r5f > gpurun_out/r5f_profile.log 2>&1
bash tools/profile_round.sh r5f32 --config 3 > gpurun_out/r5f32_profile.log 2>&1
bash tools/pmc_window.sh r5w > gpurun_out/r5w_profile.log 2>&1
{ echo "# tools/stream_probe.py (ms per 64-fit fp32 call by caller stream; 0 and 3 idle contexts created first)"; python3 tools/stream_probe.py; python3 tools/stream_probe.py --extra 3;
  echo "# tools/sweep_probe.py (cgp_sweep_fit_predict_device over one device against the plain context)"; python3 tools/sweep_probe.py; } > gpurun_out/r5_stream_groups.txt 2>&1
python3 tools/summarize_profiles.py r5f r05 > /dev/null 2>&1
python3 tools/summarize_profiles.py r5f32 r05 512 1024 f32 > /dev/null 2>&1
cp profiles/r05_kernel_stats.csv profiles/r05_kernel_stats_f32.csv profiles/r05_pmc_summary.json profiles/r05_pmc_summary_f32.json gpurun_out/ 2>/dev/null
tail -1 gpurun_out/r5f_bench.json | cut -c1-200
