#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_treplay; mkdir -p "$OUT"
# 1. the replay path on every pixel that stops (GSR_TBAND = 0.9): the whole parity suite through it
GSR_LIB=$PWD/gscream_amd/libgsraster_tband9.so GSR_SKIP_ABI_CHECK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > "$OUT/pytest_tband9.txt"
# 2. the shipped library (band 1e-4): parity suite
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_precise.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > "$OUT/pytest_shipped.txt"
# 3. full-size element-wise check per band: without the replay, band 1e-4 (shipped), band 1e-3
for v in notreplay "" tband3; do
  L=libgsraster${v:+_$v}.so
  for cfg in "1 1000000 1008 567 1 0 0" "2 1000000 1008 567 1 1 1" "3 2000000 1920 1080 1 1 1"; do
    GSR_LIB=$PWD/gscream_amd/$L GSR_SKIP_ABI_CHECK=1 timeout 900 python tools/full_size_oracle_check.py $cfg 2>>"$OUT/err.log" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pc=d['parity_check']
print('$L', '$cfg', 'px>1e-4', pc['px_gt_1e-4'], 'grad>1e-3', pc['grad_elems_gt_1e-3'], pc['grad_elems_by_cause'], 'worst', round(pc['worst_rel'],5), 'stops', pc.get('last_contributor_differs'), 'Trel', pc.get('final_T_max_rel_where_same_stop'), 'risk', pc['pixels_at_risk'])" >> "$OUT/fullsize.txt"
  done
done
# 4. what it costs
TAG=r5_treplay WORKLOADS="config2 config4" REPEAT=2 bash tools/gpu_ab.sh notreplay tband3 > /dev/null 2>&1
cat "$OUT/pytest_tband9.txt" "$OUT/pytest_shipped.txt" "$OUT/fullsize.txt" "$OUT/ab.txt"
