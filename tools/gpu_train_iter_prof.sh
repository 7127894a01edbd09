#!/bin/bash
# usage (GPU box): bash tools/gpu_train_iter_prof.sh <tag>  -- rocprofv3 kernel stats + PMC of the train-iteration row alone
set -u
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ti_$TAG; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o n -- python tools/train_iteration_probe.py > "$OUT/kt.log" 2>&1
tail -1 "$OUT/kt.log" | cut -c1-1500
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" | head -40 | cut -c1-140 | tee "$OUT/kernel_stats.md"
python tools/rocprof_durations.py "$DB" blend_fwd
python - "$DB" <<PYEOF
import sqlite3,sys
cur=sqlite3.connect(sys.argv[1]).cursor()
print([d[0] for d in cur.execute("select * from top_kernels limit 1").description])
PYEOF
rm -rf "$OUT/kt"
if [ "${2:-}" = "pmc" ]; then
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d "$OUT/p1" -o p -- python tools/train_iteration_probe.py > "$OUT/p1.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d "$OUT/p2" -o p -- python tools/train_iteration_probe.py > "$OUT/p2.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/p3" -o p -- python tools/train_iteration_probe.py > "$OUT/p3.log" 2>&1
  for d in p1 p2 p3; do DB=$(find "$OUT/$d" -name "*_results.db" | head -1); python tools/rocprof_summary.py "$DB" --pmc >> "$OUT/pmc.md"; done
  rm -rf "$OUT/p1" "$OUT/p2" "$OUT/p3"
  grep -E "gauss_bwd|blend_bwd|blend_fwd|sort_near|scatter|preprocess" "$OUT/pmc.md" | cut -c1-150
fi
