#!/bin/bash
# potf2_tile phase clocks: one fp64 N = 2048 fit (latency schedule) and the 64-fit fp32 call (mid-size schedule)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
CGP_LIB=$AB CGP_DBG=1024 python3 tools/phase_clock.py --config 2 --batch 1 > $O/r3_phase_single.json 2> $O/r3_phase_single.err
CGP_LIB=$AB CGP_DBG=1024 python3 tools/phase_clock.py --config 3 --batch 64 > $O/r3_phase_c3b64.json 2>> $O/r3_phase_single.err
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for f in ["r3_phase_single","r3_phase_c3b64"]:
    j=json.load(open(f"{O}/{f}.json"))
    print(f, "step_ms", round(j["step_ms"],4))
    for k in ["potf2_tile_ticks_per_tile","kind_A_tile_ticks_per_wg","diag_finish_ticks_per_wg"]:
        if k in j: print("  ",k,{n:(round(v/1000,2) if isinstance(v,float) else v) for n,v in j[k].items()})
PY
tail -3 $O/r3_phase_single.err
