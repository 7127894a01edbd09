#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s33; mkdir -p $OUT
FUZZ_SINGLE_KNOBS=1 FUZZ_ONLY=49 FUZZ_KNOBS=1 FUZZ_KEEP_GOING=1 timeout 900 python tools/fuzz_parity.py 60 5000 2>&1 | grep "KNOB MISMATCH.*out_color\|^case\|Error" > $OUT/fuzz49_single.txt
FUZZ_KNOBS=1 FUZZ_KEEP_GOING=1 timeout 1500 python tools/fuzz_parity.py 60 5000 2>&1 | grep "KNOB MISMATCH.*out_\|^case\|Error\|worst\|flagged" > $OUT/fuzz_all.txt
cat $OUT/fuzz49_single.txt | cut -c1-220; grep -c "^case" $OUT/fuzz_all.txt; grep "MISMATCH" $OUT/fuzz_all.txt | cut -c1-200 | head -20
