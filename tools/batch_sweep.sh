#!/bin/bash
# fits/s against the batch size (fits per call), one config: tools/batch_sweep.sh <tag> <bench args...>
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for b in 1 2 4 5 8 16 32 64 128 256 512; do
  echo -n "batch $b: "; python3 $R/bench.py --no-cpu --no-extra --batch $b --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],3), 'ms/step')"
done 2>&1 | tee $R/gpurun_out/${tag}_batchsweep.txt
