#!/bin/bash
# quick check after a kernel change: the parity / optimiser / latency tests, then the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_optimize.py -x -q -m gpu 2>&1 | tail -6 | tee $O/r3q_tests.txt
python3 bench.py --no-pmc > $O/r3q_bench.json 2> $O/r3q_bench.err
tail -1 $O/r3q_bench.json | python3 -c "
import json,sys; j=json.loads(sys.stdin.read())
print('value', j['value'], 'ms', j['ms_per_step'], 'frac', j['roofline']['frac'], 'single', j['config']['single_fit_latency_ms'])
ex=j['config']['extra']; print({k:(round(v,4) if isinstance(v,float) else v) for k,v in ex.items() if not isinstance(v,(dict,str))})
"
tail -3 $O/r3q_bench.err
