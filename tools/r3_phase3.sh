#!/bin/bash
# per-phase cycles of k_panel<float> at full batch (configs[2], 512 fits)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
CGP_LIB=$AB CGP_DBG=1024 python3 tools/phase_clock.py --config 3 --batch 512 > $O/r3_phase_c3b512.json 2> $O/r3_phase_c3b512.err
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
j=json.load(open(f"{O}/r3_phase_c3b512.json"))
print("step_ms", round(j["step_ms"],3))
for r in j["ticks_per_wg"]:
    print("  k",r["k"],"wgs",r["wgs"]," ".join(f"{n}={r[n]/1000:.1f}k" for n in ["gfetch","gram","loop","fold","wstage","trmm","store"]), "sum=%.1fk"%(sum(r[n] for n in ["gfetch","gram","loop","fold","wstage","trmm","store"])/1000))
for k in ["kind_A_tile_ticks_per_wg","diag_finish_ticks_per_wg","potf2_tile_ticks_per_tile"]:
    if k in j: print("  ",k,{n:(round(v/1000,2) if isinstance(v,float) else v) for n,v in j[k].items()})
PY
