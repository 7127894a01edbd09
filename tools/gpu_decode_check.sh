#!/bin/bash
# usage (GPU box): bash tools/gpu_decode_check.sh  -- decode parity tests, fwd / fwd+bwd timing, per-kernel averages
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/decode; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_reference_vectors2.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -12
timeout 300 python tools/decode_probe.py 2>&1 | tail -1
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o n -- python tools/decode_probe.py > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" | grep -E "^\| kernel|^\|---|gsd_" | cut -c1-120
rm -rf "$OUT/kt"
