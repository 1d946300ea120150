#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/step', round(j['roofline']['frac'],4))"; }
for i in 1 2; do
for v in gp gp_occ3 gp_fdeep; do
  echo -n "cfg3 512 lib$v: "; CGP_LIB=$R/corenav_gp_amd/libcorenav_$v.so one --config 3 --batch 512
done; done 2>&1 | tee gpurun_out/r3_ab_f32.txt
