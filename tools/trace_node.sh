#!/bin/bash
# rocprofv3 kernel-trace summary of the reference's own work item (tools/node_latency.py): which launches a callback is made of
# usage (GPU box): bash tools/trace_node.sh [tag]   -> gpurun_out/<tag>_node_stats/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r04}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_node_stats -o node -- python3 $R/tools/node_latency.py > $O/${tag}_node_stats.log 2>&1
f=$(find $O/${tag}_node_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-7 "$f" | head -12
grep -E "fixed theta|optimize" $O/${tag}_node_stats.log
