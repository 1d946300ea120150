#!/bin/bash
# Samples sclk and socket power (rocm-smi) once a second while bench.py runs 150 steps; evidence for the
# sustained-clock statement in DESIGN.md section 4.  Run through gpurun from the repo root:
#   tools/sample_clocks.sh > gpurun_out/sclk_power.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
(timeout 400 python3 $R/bench.py --no-pmc --no-cpu --steps 150 --warmup 2 > /tmp/clk_bench.json 2>/dev/null &)
echo "# t[s]  sclk  socket_power[W]   (idle samples dropped)"
for i in $(seq 1 200); do
  l=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed "s/.*: *//" | tr "\n" " ")
  case "$l" in *"(1"[0-9][0-9]"Mhz"*|*"("[0-9][0-9]"Mhz"*) ;; *) echo "$i $l";; esac
  sleep 1
  if [ -s /tmp/clk_bench.json ]; then break; fi
done
echo "# bench line:"; tail -1 /tmp/clk_bench.json | cut -c1-160
