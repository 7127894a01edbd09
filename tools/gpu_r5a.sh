#!/bin/bash
# loss kernels: four filtered planes in the SSIM forward, one-barrier reductions, depth-loss grid size
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p "$OUT"; : > "$OUT/ab.txt"
python -m pytest tests/test_gpu_loss.py -x -q -m gpu 2>&1 | tail -3 | tee "$OUT/pytest.txt"

for rep in 1 2; do
for v in "" loss5; do
  GSR_LIB=$PWD/gscream_amd/libgsraster${v:+_$v}.so python tools/rows_only.py loss depth 2>>"$OUT/err.log" | grep -v Warn | tee -a "$OUT/ab.txt"
done
done
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python tools/rows_only.py loss depth > /dev/null 2>&1
DB=$(find "$OUT/kt" -name "*_results.db" | head -1); python tools/rocprof_summary.py "$DB" | grep "gsl_\|gdl_" | cut -c1-120 | tee -a "$OUT/ab.txt"; rm -rf "$OUT/kt"
