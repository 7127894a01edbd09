#!/bin/bash
# usage (GPU box): bash tools/gpu_train_iter_timeline.sh -- GPU timeline (gaps) of the last train iteration of the probe
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ti_tl; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt" -o n -- python tools/train_iteration_probe.py > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/timeline.py "$DB" 110 130 > "$OUT/timeline.txt"
rm -rf "$OUT/kt"
cat "$OUT/timeline.txt"
