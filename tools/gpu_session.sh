#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, final library (third tier + list prefetch): the full check -- GPU tests, smoke, the driver's bench command line, the tracked
# snapshots of every workload (PMC first, bench line, rocprofv3 kernel stats), the fitted frame at 25 iterations.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_check; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
for wl in config2 config3 config4 fitted; do bash tools/snapshot.sh r06f_$wl $wl > $OUT/snap_$wl.log 2>&1; done
for wl in init_state surfaces; do timeout 900 python bench.py --workload $wl 2> $OUT/bench_$wl.err | tail -1 > $OUT/bench_$wl.json; done
GSR_FIT_ITERS=25 timeout 900 python bench.py --workload fitted --no-next-rows --no-strict-parity 2> $OUT/bench_fitted25.err | tail -1 > $OUT/bench_fitted_25.json
tail -n 3 $OUT/pytest.txt; tail -n 3 $OUT/smoke.txt
