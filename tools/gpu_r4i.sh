#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p "$OUT"; : > "$OUT/ab.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_culling_only or occlusion or banded" 2>&1 | tail -60 > "$OUT/pytest_focus.txt"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[Gloo\]" | tail -30 > "$OUT/pytest.txt"
for b in "" "--occlusion 1"; do
  timeout 600 python bench.py --no-cpu-baseline --no-strict-parity $b 2>>"$OUT/err.log" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); ti=d['next_rows']['train_iteration']; print('bench [$b]', d['value'], 'train_iteration', ti['ms_per_iteration'], ti.get("num_rendered"), ti.get("num_occluded"), {k.replace('void ','')[:28]:v for k,v in list(ti['gpu_top_kernels_us'].items())[:12]}); print('   pipeline', d['next_rows']['pipeline_decode_raster_loss']['ms_per_iteration'], 'fps', d['next_rows']['render_fps']['standin_model_view']['fps'])" | tee -a "$OUT/ab.txt"
done
