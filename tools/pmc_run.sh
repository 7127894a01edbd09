#!/bin/bash
# Collects PMC counters for bench.py in separate rocprofv3 passes (never combined with --stats / tracing).
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <workload> <outdir>
set -u
WL=${1:-config2}; OUT=${2:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-next-rows"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT" -o p1 -- $CMD > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT" -o p2 -- $CMD > "$OUT/p2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$OUT" -o p3 -- $CMD > "$OUT/p3.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o p4 -- $CMD > "$OUT/p4.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o p5 -- $CMD > "$OUT/p5.log" 2>&1
ls -la "$OUT"
