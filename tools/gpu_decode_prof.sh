#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/decode; mkdir -p "$OUT"
python tools/decode_probe.py 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python tools/decode_probe.py > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" | cut -c1-150 | head -40
rm -rf "$OUT/kt"
