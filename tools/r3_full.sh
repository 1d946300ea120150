#!/bin/bash
# full GPU suite + the default bench line (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/r3_gputests.txt
python3 bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err
tail -1 $O/r3_bench_default.json | python3 -c "
import json,sys; j=json.loads(sys.stdin.read())
print('value', j['value'], 'ms', j['ms_per_step'], 'frac', j['roofline']['frac'], 'single', j['config']['single_fit_latency_ms'])
ex=j['config']['extra']; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in ex.items() if not isinstance(v,(dict,str))})
print(j['cpu_baseline']['value'], j['cpu_baseline']['sample'][:200]); print(j['cpu_baseline'].get('port_threads_over_batch'))
"
tail -5 $O/r3_bench_default.err
