#!/bin/bash
# small windows (N = 256 and N = 128): the throughput schedule (variant: make variant NAME=nt3 EXTRA=-DCGP_LAT_MIN_NT=3, the setting
# until the end of round 3) against the latency schedule (shipped) by call size
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), end=' ')"; }
for n in 256 128; do for dt in f64 f32; do
for b in 1 2 4 8 11 16 20; do
  echo -n "N=$n $dt batch $b ms/call [throughput, latency]: "
  if [ $dt = f64 ]; then cfg=1; else cfg=3; fi
  CGP_LIB=$R/corenav_gp_amd/libcorenav_gp_nt3.so one --config $cfg --n $n --batch $b
  CGP_LIB=$R/corenav_gp_amd/libcorenav_gp.so one --config $cfg --n $n --batch $b; echo
done; done; done 2>&1 | tee gpurun_out/r3_nt_sweep.txt
