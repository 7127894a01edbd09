#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, final source (commit cf36f6c + recorded experiments): PMC + bench + kernel stats of every BASELINE config and the fitted frame,
# then the driver's bench command line (PMC status must read "current").
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_check3; mkdir -p $OUT
for wl in config2 config3 config4 fitted; do bash tools/snapshot.sh r06h_$wl $wl > $OUT/snap_$wl.log 2>&1; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final_check3/bench_driver_style.json"))
print("driver-style", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_source",{}).get("status"))
for wl in ("config2","config3","config4","fitted"):
    s=json.load(open(f"gpurun_out/snap_r06h_{wl}/bench.json")); print(wl, s["value"], s["ms_per_step"], s.get("ms_per_step_spread",{}).get("blocks_ms"), s["roofline"]["frac"])
PY
