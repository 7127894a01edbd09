#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_s28; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -6 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
for wl in config2 config3 config4 init_state surfaces; do timeout 900 python bench.py --workload $wl 2> $OUT/bench_$wl.err | tail -1 > $OUT/bench_$wl.json; done
GSR_LIB=$PWD/gscream_amd/libgsraster_trace.so GSR_SKIP_ABI_CHECK=1 timeout 600 python tools/wave_trace.py config2 fwd > $OUT/trace_config2_fwd.txt 2>&1
tail -3 $OUT/pytest.txt $OUT/smoke.txt
