#!/bin/bash
# Round-3: mid-size schedule (kind C + deep-prefetch fp32): parity first, then the call-size sweep with the
# crossover forced either side (ablation build, CGP_MID_FITS).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mid_size or as_sharded_64 or golden_fp64 or fp32_full_size or config2_full_size_throughput or ragged_batches or stream_groups" 2>&1 | tail -15 | tee $O/r3_mid_tests.txt
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/step')"; }
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
{
for b in 32 64 96 128 192 256 512; do
  echo -n "cfg3 batch $b shipped: "; one --config 3 --batch $b
  echo -n "cfg3 batch $b ab mid=0: "; CGP_LIB=$AB CGP_MID_FITS=0 one --config 3 --batch $b
  echo -n "cfg3 batch $b ab mid=512: "; CGP_LIB=$AB CGP_MID_FITS=512 one --config 3 --batch $b
done
for b in 32 64 128 256; do
  echo -n "cfg2 batch $b ab mid=0: "; CGP_LIB=$AB CGP_MID_FITS=0 one --config 2 --batch $b
  echo -n "cfg2 batch $b ab mid=512: "; CGP_LIB=$AB CGP_MID_FITS=512 one --config 2 --batch $b
done
} 2>&1 | tee $O/r3_mid_sweep.txt
CGP_PROF_DUMP=1 python3 bench.py --no-pmc --no-cpu --no-extra --steps 2 --warmup 1 --config 3 --batch 64 2>&1 >/dev/null | grep "cgp prof" | tail -12 > $O/r3_mid_dump64.txt
cat $O/r3_mid_dump64.txt
