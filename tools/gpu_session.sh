#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s7; mkdir -p "$OUT"
timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\[Gloo\]" | grep -E "passed|failed|rror|assert|threshold flips" | tail -30 > "$OUT/pytest_all.txt"
TAG=r5_s7 WORKLOADS="config2 config3 config4" REPEAT=2 bash tools/gpu_ab.sh notreplay > /dev/null 2>&1
bash tools/snapshot.sh r05_config2 config2 > "$OUT/snap2.log" 2>&1
bash tools/snapshot.sh r05_config3 config3 > "$OUT/snap3.log" 2>&1
bash tools/snapshot.sh r05_config4 config4 > "$OUT/snap4.log" 2>&1
for wl in surfaces init_state; do timeout 900 python bench.py --workload $wl --no-strict-parity 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_$wl.json"; done
timeout 900 python bench.py --workload config5 --steps 50 --warmup 10 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_config5_1gpu.json"
cat "$OUT/pytest_all.txt" "$OUT/ab.txt"
python - <<'EOF'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for tag in ("r05_config2", "r05_config3", "r05_config4"):
    try:
        d = json.load(open(o + f"snap_{tag}/bench.json"))
        pc = d.get("parity_check", {})
        print(tag, d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "whole", d["whole_iteration"]["frac_of_hbm_peak"],
              "parity px", pc.get("px_gt_1e-4"), "grad", pc.get("grad_elems_gt_1e-3"), pc.get("grad_elems_by_cause"), pc.get("last_contributor_differs"),
              "env", (pc.get("order_noise_envelope") or {}).get("elements_inside"), (pc.get("order_noise_envelope") or {}).get("elements_outside"))
    except Exception as ex:
        print(tag, "error", ex)
for wl in ("surfaces", "init_state", "config5_1gpu"):
    try:
        d = json.load(open(o + f"r5_s7/bench_{wl}.json"))
        print(wl, d["value"], d["ms_per_step"], d.get("wall_clock"), {k: round(x["avg_ms"] * 1e3, 1) for k, x in d.get("stages", {}).items()})
    except Exception as ex:
        print(wl, "error", ex)
EOF
