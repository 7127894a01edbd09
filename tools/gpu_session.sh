#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, final library (segment layout 7 x L / 6 x 3 L / 13 x 8 L, deep-first flag, list prefetch): smoke, the tracked snapshots of every
# workload (PMC first, bench line, rocprofv3 kernel stats), the driver's bench command line, the fitted frame along the run.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_check2; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
for wl in config2 config3 config4 fitted; do bash tools/snapshot.sh r06g_$wl $wl > $OUT/snap_$wl.log 2>&1; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
for wl in init_state surfaces; do timeout 900 python bench.py --workload $wl 2> $OUT/bench_$wl.err | tail -1 > $OUT/bench_$wl.json; done
for it in 25 100 1600; do GSR_FIT_ITERS=$it timeout 900 python bench.py --workload fitted --no-next-rows --no-strict-parity --cpu-budget 6 2> $OUT/bench_fitted$it.err | tail -1 > $OUT/bench_fitted_$it.json; done
tail -n 2 $OUT/smoke.txt
