#!/bin/bash
# Round 6, final library: second randomised sweep again (120 cases, seed 2000, every knob) with the T-stop tie classified; the parity suites
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/fuzz120; mkdir -p $OUT
FUZZ_KNOBS=1 timeout 2400 python tools/fuzz_parity.py 120 2000 > $OUT/fuzz.txt 2>&1
tail -n 4 $OUT/fuzz.txt | cut -c1-300; grep -c " ok" $OUT/fuzz.txt; grep "classified" $OUT/fuzz.txt | head -5 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -3
