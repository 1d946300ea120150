#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
AB=$R/corenav_gp_amd/libcorenav_gp_ab.so
for b in 16 64; do
CGP_SCHED=throughput CGP_LIB=$AB CGP_DBG=1024 python3 tools/phase_clock.py --config 3 --batch $b > $O/r3_phase_mid$b.json 2> $O/r3_phase_mid$b.err
CGP_SCHED=throughput CGP_LIB=$AB CGP_DBG=1024 CGP_MID_FITS=0 python3 tools/phase_clock.py --config 3 --batch $b > $O/r3_phase_nomid$b.json 2>> $O/r3_phase_mid$b.err
done
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for f in ["r3_phase_mid16","r3_phase_nomid16","r3_phase_mid64","r3_phase_nomid64"]:
    j=json.load(open(f"{O}/{f}.json"))
    print(f, "step_ms", round(j["step_ms"],3))
    for r in j["ticks_per_wg"]:
        print("  k",r["k"],"wgs",r["wgs"]," ".join(f"{n}={r[n]/1000:.1f}k" for n in ["gfetch","gram","loop","fold","wstage","trmm","store"]))
PY
