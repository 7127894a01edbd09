#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p "$OUT"; : > "$OUT/ab.txt"
cat > /tmp/row.py <<'P'
import json,sys
d=json.loads(sys.stdin.read()); ti=d['next_rows']['train_iteration']
print(sys.argv[1], 'train_iteration', ti['ms_per_iteration'], 'kernels', ti['gpu_kernel_ms_sum'], {k.replace('void ','')[:28]:v for k,v in list(ti['gpu_top_kernels_us'].items())[:14] if 'gauss' in k})
P
for v in "" skipsum; do for occ in -1 0; do
  GSR_LIB=$PWD/gscream_amd/libgsraster${v:+_$v}.so timeout 600 python bench.py --no-cpu-baseline --no-strict-parity --occlusion $occ 2>>"$OUT/err.log" | tail -1 | python /tmp/row.py "${v:-cur} occ=$occ" | tee -a "$OUT/ab.txt"
done; done
