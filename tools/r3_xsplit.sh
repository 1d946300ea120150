#!/bin/bash
# Round-3: mid-size calls with the extra rows on a second stream (factorisation || extra rows) against the variant
# without (-DCGP_NO_EXTRA_SPLIT=1): parity first, then ms per call on one and two contexts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mid_size or as_sharded_64 or fp32_full_size or config2_full_size_throughput or ragged_batches or stream_groups or default_stream or batch_jitter or throughput_schedule_matches" 2>&1 | tail -8 | tee $O/r3_xsplit_tests.txt
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/call', end='')"; }
NOX=$R/corenav_gp_amd/libcorenav_gp_nox.so
{
for b in 24 32 48 64 96; do
  echo -n "cfg3 batch $b split: "; one --config 3 --batch $b --pipeline 1; echo -n " | two ctx: "; one --config 3 --batch $b --pipeline 2; echo -n " || nosplit: "; CGP_LIB=$NOX one --config 3 --batch $b --pipeline 1; echo -n " | two ctx: "; CGP_LIB=$NOX one --config 3 --batch $b --pipeline 2; echo
done
for b in 12 24 32 48; do
  echo -n "cfg2 batch $b split: "; one --config 2 --batch $b --pipeline 1; echo -n " | two ctx: "; one --config 2 --batch $b --pipeline 2; echo -n " || nosplit: "; CGP_LIB=$NOX one --config 2 --batch $b --pipeline 1; echo
done
} 2>&1 | tee $O/r3_xsplit_sweep.txt
CGP_PROF_DUMP=1 python3 bench.py --no-pmc --no-cpu --no-extra --steps 2 --warmup 1 --config 3 --batch 64 --pipeline 1 2>&1 >/dev/null | grep "cgp prof" | tail -20 > $O/r3_xsplit_dump64.txt
