#!/bin/bash
# A/B of two builds of libcorenav_gp.so on the same box: tools/ab_libs.sh <old.so> <bench args...>
# (alternates new/old three times; prints fits/s and ms per step)
old=$1; shift
lib=corenav_gp_amd/libcorenav_gp.so
cp $lib /tmp/new.so
for i in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/new.so $lib; else cp $old $lib; fi
    echo -n "$v: "; timeout 300 python bench.py --no-pmc --no-cpu "$@" | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['ms_per_step'],3), j['kernel_ms_per_step'])"
  done
done
cp /tmp/new.so $lib
