#!/bin/bash
# fits/s against the call size (fits per call), one config: tools/batch_sweep.sh <tag> <bench args...>
#   column 1: calls back to back on ONE context (ms per call = the call's own duration);
#   column 2: the same calls dealt over TWO contexts on two HIP streams (bench.py --pipeline 2: successive calls overlap)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
one() { python3 $R/bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/call', end='')"; }
for b in 1 2 4 8 16 24 32 48 64 96 128 256 512; do
  echo -n "batch $b: "; one --batch $b --pipeline 1 "$@"; echo -n "   | two contexts: "; one --batch $b --pipeline 2 "$@"; echo
done 2>&1 | tee $R/gpurun_out/${tag}_batchsweep.txt
