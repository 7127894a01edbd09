#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6: forward pair loop unrolled (GSR_FWD_UNROLL) / fewer waves per SIMD -- does a lone deep walk get faster?  (A/B, not shipped)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WORKLOADS="init_state config2 fitted" STEPS=40 WARMUP=10 TAG=fwd_unroll bash tools/gpu_ab.sh u2 u2w5 w5 u4w4
