#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/batch_sweep.sh r3f32 --config 3 > /dev/null
bash tools/profile_round.sh r3d > /dev/null
bash tools/profile_round.sh r3c --config 3 > /dev/null
tail -3 gpurun_out/r3f32_batchsweep.txt
