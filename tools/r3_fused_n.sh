#!/bin/bash
# fp64 throughput schedule: diagonal tiles inside the panel launches (fuseddiag) against one k_diag_lean launch per step (splitdiag),
# by window length and call size (ablation build), ms per call
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export CGP_LIB=$R/corenav_gp_amd/libcorenav_gp_ab.so
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), end=' ')"; }
for n in 256 512 1024 2048; do for b in 128 256 512 1024; do
  echo -n "N=$n batch $b ms/call [split, fused]: "
  CGP_SCHED=splitdiag one --config 2 --n $n --batch $b; CGP_SCHED=fuseddiag one --config 2 --n $n --batch $b; echo
done; done 2>&1 | tee gpurun_out/r3_fused_n.txt
