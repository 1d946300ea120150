#!/bin/bash
# extra rows on a second stream (fp64 mid-size calls from 28 fits): with (shipped) / without (variant noxs), by window length
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 4 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), end=' ')"; }
for n in 512 768 1024 1536 2048; do for b in 28 36 48; do
  echo -n "N=$n batch $b ms/call [split, no split]: "
  CGP_SCHED=throughput CGP_LIB=$R/corenav_gp_amd/libcorenav_gp.so one --config 2 --n $n --batch $b
  CGP_SCHED=throughput CGP_LIB=$R/corenav_gp_amd/libcorenav_gp_noxs.so one --config 2 --n $n --batch $b; echo
done; done 2>&1 | tee gpurun_out/r3_xs_n.txt
