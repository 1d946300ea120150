#!/bin/bash
# Round-3 evidence from ONE box: call-size sweep, bench lines, rocprofv3 kernel-trace stats and PMC passes of the default
# command and of --config 3 (profile_round.sh), condensed afterwards with tools/summarize_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r3d > /dev/null
bash tools/profile_round.sh r3c --config 3 > /dev/null
bash tools/batch_sweep.sh r3f32 --config 3 > /dev/null
tail -3 gpurun_out/r3f32_batchsweep.txt
