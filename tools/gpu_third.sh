#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/third; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_issue tools/microbench/valu_issue.hip && timeout 600 /tmp/valu_issue > "$OUT/valu_issue.md" 2>&1
grep -E "\| 8 \|" "$OUT/valu_issue.md"
bash tools/snapshot.sh r02a > "$OUT/snapshot.log" 2>&1
cat gpurun_out/snap_r02a/kernel_stats.md | head -30
tail -12 gpurun_out/snap_r02a/pmc.md
