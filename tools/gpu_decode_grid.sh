#!/bin/bash
# usage (GPU box): bash tools/gpu_decode_grid.sh  -- decode tests + emit time for a few persistent grid sizes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/decode; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_reference_vectors2.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -5
for G in 128 256 384 512 768; do
  GSD_EMIT_GRID=$G timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o n -- python tools/decode_probe.py > "$OUT/kt.log" 2>&1
  DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
  echo "grid $G: $(python tools/rocprof_summary.py "$DB" | grep -E "gsd_emit|gsd_count" | cut -c1-80 | tr '\n' ' ')"
  rm -rf "$OUT/kt"
done
