#!/bin/bash
# round 4, step 1: guard band -- band frequency, full-size classification with / without the band, A/B, parity suites
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p "$OUT"
GSR_LIB=$PWD/gscream_amd/libgsraster_count.so timeout 300 python tools/blend_counts.py config2 > "$OUT/counts.txt" 2>&1
for v in "" noband; do
  L=libgsraster${v:+_$v}.so
  GSR_LIB=$PWD/gscream_amd/$L timeout 600 python tools/full_size_oracle_check.py 1 1000000 1008 567 1 0 0 > "$OUT/full_c2_${v:-band}.json" 2>>"$OUT/err.log"
  GSR_LIB=$PWD/gscream_amd/$L timeout 600 python tools/full_size_oracle_check.py 2 1000000 1008 567 1 1 1 > "$OUT/full_c3_${v:-band}.json" 2>>"$OUT/err.log"
  GSR_LIB=$PWD/gscream_amd/$L timeout 900 python tools/full_size_oracle_check.py 3 2000000 1920 1080 1 1 1 > "$OUT/full_c4_${v:-band}.json" 2>>"$OUT/err.log"
done
bash tools/gpu_ab2.sh noband > "$OUT/ab.txt" 2>&1
timeout 1200 python -m pytest tests/test_gpu_precise.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -5 > "$OUT/pytest2.txt"
timeout 600 python bench.py --steps 50 --warmup 10 --no-next-rows 2>>"$OUT/err.log" | tail -1 > "$OUT/bench.json"
