#!/bin/bash
# usage (GPU box): bash tools/gpu_pipeline_prof.sh  -- rocprofv3 kernel stats of the decode -> raster -> loss pipeline row alone
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pipeline; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o n -- python tools/pipeline_probe.py > "$OUT/kt.log" 2>&1
tail -1 "$OUT/kt.log" | cut -c1-300
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" | head -32 | cut -c1-120 | tee "$OUT/kernel_stats.md"
rm -rf "$OUT/kt"
