#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/fuzz_final; mkdir -p $OUT
FUZZ_KNOBS=1 timeout 1500 python tools/fuzz_parity.py 60 > $OUT/fuzz.txt 2>&1
tail -n 5 $OUT/fuzz.txt | cut -c1-300; grep -c " ok" $OUT/fuzz.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
