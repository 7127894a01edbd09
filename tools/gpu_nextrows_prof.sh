#!/bin/bash
# rocprofv3 kernel stats of the SURVEY 8(f) rows (full bench.py line incl. next_rows) -> gpurun_out/nextrows/kernel_stats.md
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/nextrows; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o n -- python bench.py --no-cpu-baseline --no-strict-parity --steps 20 > "$OUT/bench.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" | grep -E "^\| kernel|^\|---|gsl_|gdl_|gsd_|gsk_|gst_|rocprim" > "$OUT/kernel_stats.md"
rm -rf "$OUT/kt"
cat "$OUT/kernel_stats.md" | cut -c1-140
