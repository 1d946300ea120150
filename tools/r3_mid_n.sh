#!/bin/bash
# the mid-size form of the throughput schedule (kinds C / image-A, fp32 deep loops + fat tile) by window length:
# ablation build, CGP_MID_FITS = 0 (off) / 512 (on), ms per call
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export CGP_LIB=$R/corenav_gp_amd/libcorenav_gp_ab.so CGP_LAT_FITS=0
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 4 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), end=' ')"; }
for cfg in 2 3; do for n in 256 512 1024; do for b in 24 48 96 160; do
  echo -n "config $cfg N=$n batch $b ms/call [mid off, mid on]: "
  CGP_MID_FITS=0 one --config $cfg --n $n --batch $b; CGP_MID_FITS=512 one --config $cfg --n $n --batch $b; echo
done; done; done 2>&1 | tee gpurun_out/r3_mid_n.txt
