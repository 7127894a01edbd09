#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p "$OUT"
bash tools/gpu_ab_r3.sh > "$OUT/ab_log.txt" 2>&1
cp gpurun_out/ab_r3/ab.txt "$OUT/ab.txt"; cp gpurun_out/ab_r3/pytest.txt "$OUT/pytest.txt"
bash tools/gpu_train_iter_prof.sh r04 > "$OUT/ti_prof.txt" 2>&1
