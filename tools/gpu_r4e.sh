#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p "$OUT"
timeout 600 python tools/occlusion_cull_estimate.py large > "$OUT/occl_large.txt" 2>&1
timeout 600 python tools/occlusion_cull_estimate.py bench > "$OUT/occl_bench.txt" 2>&1
bash tools/gpu_ab_r3.sh nobandfwd > "$OUT/ab_log.txt" 2>&1
cp gpurun_out/ab_r3/ab.txt "$OUT/ab.txt"; cp gpurun_out/ab_r3/pytest.txt "$OUT/pytest.txt"
