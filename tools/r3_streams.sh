#!/bin/bash
# headline (512 x N=2048 fp64, split diagonal launches): one stream against two / four stream groups, and two contexts
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],3), 'ms', end='')"; }
for rep in 1 2; do
for st in 1 2 4; do echo -n "streams $st: "; one --streams $st; echo; done
echo -n "two contexts (pipeline 2): "; one --pipeline 2; echo
echo -n "batch 1024 one stream: "; one --batch 1024; echo
done 2>&1 | tee gpurun_out/r3_streams.txt
