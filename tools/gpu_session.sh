#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=r5_s39 WORKLOADS="config2 config3 config4 init_state" REPEAT=2 bash tools/gpu_ab.sh prio16 prio32
