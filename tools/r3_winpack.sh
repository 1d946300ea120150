#!/bin/bash
# Round-3 sliding window: paired ticks with 1 / 2 / 4 windows per workgroup (CGP_WIN_WPW, ablation build): parity, then ticks/s
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
for w in 1 2 4; do
  echo "== pairs, WPW $w"
  CGP_LIB=$AB CGP_WIN_WPW=$w timeout 900 python3 -m pytest tests/test_gpu_window.py -x -q -m gpu 2>&1 | tail -1
  CGP_LIB=$AB CGP_WIN_WPW=$w python3 tests/fuzz/fuzz_window.py 30 3 2>&1 | tail -1
  for nw in 256 512 1024 2048 4096; do CGP_LIB=$AB CGP_WIN_WPW=$w python3 tools/bench_window.py --windows $nw --ticks 200 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('windows', j['windows'], round(j['value']/1e6,3), 'M ticks/s  frac', round(j['frac'],3), 'us/tick', round(j['us_per_tick_per_window'],1), j['info'], round(j['logml_last'],6))"; done
done 2>&1 | tee gpurun_out/r3_winpack.txt
