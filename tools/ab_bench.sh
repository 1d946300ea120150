#!/bin/bash
# One parameterised same-box A/B (replaces round 3's two dozen one-off r3_*.sh wrappers; their results live in profiles/r03_*.txt):
#   tools/ab_bench.sh [-r rounds] [-l "lib1 lib2 ..."] [-e "ENV1=a ENV1=b ..."] -- <bench.py args ...>
# runs `bench.py --no-pmc --no-cpu --no-extra <args>` for every library (names under corenav_gp_amd/: gp, gp_ab, a `make variant`
# NAME as gp_<NAME>) x every environment setting, alternating, `rounds` times (default 2), and prints fits/s, ms per step and
# the k_panel roofline fraction per run.  Examples (each was a script of its own in round 3):
#   tools/ab_bench.sh -l "gp gp_occ3 gp_fdeep" -- --config 3 --batch 512                      # variant builds of the fp32 full-batch kernel
#   tools/ab_bench.sh -l gp_ab -e "CGP_MID_FITS=0 CGP_MID_FITS=512" -- --config 2 --n 1536 --batch 96   # mid-size form off / on
#   tools/ab_bench.sh -l gp_ab -e "CGP_LAT_FITS=0 CGP_LAT_FITS=64" -- --config 3 --batch 20              # latency | throughput crossover
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
rounds=2; libs="gp"; envs="_=_"
while getopts "r:l:e:" o; do case $o in r) rounds=$OPTARG;; l) libs=$OPTARG;; e) envs=$OPTARG;; esac; done
shift $((OPTIND - 1)); [ "$1" = "--" ] && shift
for i in $(seq $rounds); do for l in $libs; do for e in $envs; do
  echo -n "lib$l $e: "
  env CGP_LIB=$R/corenav_gp_amd/libcorenav_$l.so $e python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 |
    python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/step', round(j['roofline']['frac'],4))"
done; done; done
