#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, final library: the bench lines again on a fresh box (the snapshot's config-2 region caught a slow block: 0.403 / 0.404 / 0.518)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_lines; mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
timeout 900 python bench.py 2> $OUT/bench2.err | tail -1 > $OUT/bench_config2.json
python - <<'PY'
import json
for f in ("bench_driver_style","bench_config2"):
    d=json.load(open(f"gpurun_out/final_lines/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d.get("ms_per_step_spread",{}).get("blocks_ms"), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source",{}).get("status"), d.get("view_cache",{}).get("ms_per_step_without"), (d.get("rotation") or {}).get("ms_per_step"))
PY
