#!/bin/bash
# Where does the latency schedule stop paying?  fits/s against the batch with the crossover forced either side
# (ablation build, CGP_LAT_FITS): tools/lat_crossover.sh <tag> <bench args...>
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export CGP_LIB=$R/corenav_gp_amd/libcorenav_gp_ab.so
for b in 4 8 12 16 20 24 28 32 48; do
  for lf in 0 64; do
    echo -n "batch $b lat_fits $lf: "
    CGP_LAT_FITS=$lf python3 $R/bench.py --no-pmc --no-cpu --no-extra --batch $b --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],3), 'ms/step')"
  done
done 2>&1 | tee $R/gpurun_out/${tag}_latcross.txt
