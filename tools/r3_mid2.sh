#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/step')"; }
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
{
for b in 32 48 64 96 128 160 192 256; do
  echo -n "cfg3 batch $b ab mid=0: "; CGP_LIB=$AB CGP_MID_FITS=0 one --config 3 --batch $b
  echo -n "cfg3 batch $b ab mid=512: "; CGP_LIB=$AB CGP_MID_FITS=512 one --config 3 --batch $b
done
for b in 24 32 48 64 96; do
  echo -n "cfg2 batch $b ab mid=0: "; CGP_LIB=$AB CGP_MID_FITS=0 one --config 2 --batch $b
  echo -n "cfg2 batch $b ab mid=512: "; CGP_LIB=$AB CGP_MID_FITS=512 one --config 2 --batch $b
done
} 2>&1 | tee $O/r3_mid2_sweep.txt
CGP_PROF_DUMP=1 python3 bench.py --no-pmc --no-cpu --no-extra --steps 2 --warmup 1 --config 3 --batch 64 2>&1 >/dev/null | grep "cgp prof" | tail -12 > $O/r3_mid2_dump64.txt
cat $O/r3_mid2_dump64.txt
