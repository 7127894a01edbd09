#!/bin/bash
# round 4: banded scatter -- tests, config 2/3/4 + large-splat rows, A/B against the round-3 library
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p "$OUT"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_decode.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -6 > "$OUT/pytest.txt"
bash tools/gpu_ab_r3.sh > "$OUT/ab_log.txt" 2>&1
cp gpurun_out/ab_r3/ab.txt "$OUT/ab.txt"
timeout 900 python bench.py --no-cpu-baseline --no-strict-parity 2>>"$OUT/err.log" | tail -1 > "$OUT/bench.json"
