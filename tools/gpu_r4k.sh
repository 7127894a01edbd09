#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p "$OUT"
bash tools/gpu_ab_r3.sh > "$OUT/ab_log.txt" 2>&1
cp gpurun_out/ab_r3/ab.txt "$OUT/ab.txt"; cp gpurun_out/ab_r3/pytest.txt "$OUT/pytest.txt"
for b in "-1" "0"; do
  timeout 600 python bench.py --no-cpu-baseline --no-strict-parity --occlusion $b 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_occ$b.json"
done
