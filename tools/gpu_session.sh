#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s6; mkdir -p "$OUT"
timeout 3000 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^\[Gloo\]" | grep -E "passed|failed|error|Error|assert|\[shipped\]|\[precise\]|threshold flips" | tail -40 > "$OUT/pytest_all.txt"
TAG=r5_s6 WORKLOADS="config2 config3 config4" REPEAT=2 bash tools/gpu_ab.sh notreplay > /dev/null 2>&1
cat "$OUT/pytest_all.txt" "$OUT/ab.txt"
