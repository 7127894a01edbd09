#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, final library: snapshots (PMC, bench line, rocprofv3 kernel stats) of the two remaining probe workloads, init_state and surfaces
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_check5; mkdir -p $OUT
for wl in init_state surfaces; do bash tools/snapshot.sh r06j_$wl $wl > $OUT/snap_$wl.log 2>&1; done
python - <<'PY'
import json
for wl in ("init_state","surfaces"):
    s=json.load(open(f"gpurun_out/snap_r06j_{wl}/bench.json")); pc=s.get("parity_check") or {}
    print(wl, s["value"], s["ms_per_step"], s.get("ms_per_step_spread",{}).get("blocks_ms"), s["roofline"]["frac"], pc.get("px_gt_1e-4"), pc.get("grad_elems_gt_1e-3"))
PY
