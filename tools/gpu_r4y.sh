#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4y; mkdir -p "$OUT"; : > "$OUT/ab.txt"
for rep in 1 2; do
for v in "" dnoslp dilp dmaxilp lilp; do
  GSR_LIB=$PWD/gscream_amd/libgsraster${v:+_$v}.so python tools/rows_only.py decode loss 2>>"$OUT/err.log" | grep -v Warn | tee -a "$OUT/ab.txt"
done
done
