#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s37; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walks_into_the_second_tier" 2>&1 | grep -v "^\[Gloo\]" | tail -40 > $OUT/pytest.txt
cat $OUT/pytest.txt
