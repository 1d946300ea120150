#!/bin/bash
# Round-3 diagnostics of the mid-size fp32 call (64 x N=1024, configs[2] per GPU at 8 GPUs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
one() { python3 bench.py --no-pmc --no-cpu --no-extra --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value'],1), 'fits/s', round(j['ms_per_step'],4), 'ms/step')"; }
{
for b in 32 64 128; do for s in 1 2 4 8; do echo -n "cfg3 batch $b streams $s: "; one --config 3 --batch $b --streams $s; done; done
for b in 64; do for s in 1 2 4; do echo -n "cfg2 batch $b streams $s: "; one --config 2 --batch $b --streams $s; done; done
} 2>&1 | tee $O/r3_streams.txt
CGP_PROF_DUMP=1 python3 bench.py --no-pmc --no-cpu --no-extra --steps 2 --warmup 1 --config 3 --batch 64 2>&1 >/dev/null | grep "cgp prof" | tail -24 > $O/r3_dump64.txt
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
CGP_LIB=$AB CGP_DBG=1024 python3 tools/phase_clock.py --config 3 --batch 64 > $O/r3_phase64.json 2> $O/r3_phase64.err
CGP_LIB=$AB CGP_DBG=3072 python3 tools/launch_spans.py --config 3 --batch 64 > $O/r3_spans64.json 2> $O/r3_spans64.err
CGP_LIB=$AB CGP_DBG=3072 python3 tools/launch_spans.py --config 3 --batch 16 > $O/r3_spans16.json 2> $O/r3_spans16.err
cat $O/r3_dump64.txt | tail -12
