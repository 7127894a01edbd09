#!/bin/bash
# round-4 evidence, second half of the round (two-level sort, no-SLP blend / per-Gaussian backward): everything profiles/r04_* holds
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4final; mkdir -p "$OUT"
bash tools/gpu_r4_final.sh
for wl in config2 config3 config4; do python bench.py --workload $wl --no-next-rows 2>/dev/null | tail -1 > "$OUT/bench_$wl.json"; done
python bench.py 2>/dev/null | tail -1 > "$OUT/bench_full.json"
python bench.py --workload surfaces --no-next-rows --no-strict-parity 2>/dev/null | tail -1 > "$OUT/bench_surfaces.json"
bash tools/gpu_train_iter_prof.sh r4 > "$OUT/ti_prof.txt" 2>&1; cp gpurun_out/ti_r4/kernel_stats.md "$OUT/train_iteration_kernel_stats.md"
FUZZ_KNOBS=1 timeout 900 python tools/fuzz_parity.py 64 4000 > "$OUT/fuzz.txt" 2>&1
bash tools/gpu_ab_r3.sh > /dev/null 2>&1; cp gpurun_out/ab_r3/ab.txt "$OUT/ab_r3.txt"
