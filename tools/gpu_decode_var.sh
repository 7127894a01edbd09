#!/bin/bash
# usage (GPU box): bash tools/gpu_decode_var.sh VARIANT...  -- per-kernel decode times of gscream_amd/libgsraster_<VARIANT>.so (diagnostic builds)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/decode; mkdir -p "$OUT"
one() {
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o n -- python tools/decode_probe.py > "$OUT/kt.log" 2>&1
  DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
  echo "$1: $(python tools/rocprof_summary.py "$DB" | grep -E "$PATTERN" | awk -F'|' '{printf "%s %s | ", $2, $5}')"
  rm -rf "$OUT/kt"
}
PATTERN=${PATTERN:-gsd_}
one base
for v in "$@"; do export GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so; one $v; done
