#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_s27; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "view_cache" 2>&1 | tail -5 > $OUT/pytest.txt
for wl in config2 config3 surfaces init_state; do for k in bwd fwd; do
  GSR_LIB=$PWD/gscream_amd/libgsraster_trace.so GSR_SKIP_ABI_CHECK=1 timeout 600 python tools/wave_trace.py $wl $k > $OUT/trace_${wl}_$k.txt 2>&1
done; done
tail -3 $OUT/pytest.txt
