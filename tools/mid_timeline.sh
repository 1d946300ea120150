#!/bin/bash
# rocprofv3 kernel trace of a mid-size call, condensed into a per-launch timeline of the LAST call of the run:
#   bash tools/mid_timeline.sh <tag> [mid_call.py args]   -> gpurun_out/<tag>_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-mid}; shift || true
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_trace -o t -- python3 $R/tools/mid_call.py --reps 30 "$@" > $O/${tag}_trace.log 2>&1
f=$(find $O/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/mid_timeline.py "$f" > $O/${tag}_timeline.txt
rm -rf $O/${tag}_trace
cat $O/${tag}_timeline.txt
tail -1 $O/${tag}_trace.log
