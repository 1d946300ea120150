#!/bin/bash
# fp64 mid-size form between 48 and 96 fits per call by window length (ablation build, CGP_MID_FITS 0 / 512), ms per call
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export CGP_LIB=$R/corenav_gp_amd/libcorenav_gp_ab.so CGP_LAT_FITS=0
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 4 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), end=' ')"; }
for n in 768 1024 1536 2048; do for b in 56 64 80 96; do
  echo -n "N=$n batch $b ms/call [mid off, mid on]: "
  CGP_MID_FITS=0 one --config 2 --n $n --batch $b; CGP_MID_FITS=512 one --config 2 --n $n --batch $b; echo
done; done 2>&1 | tee gpurun_out/r3_mid_n2.txt
