#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, end of the second session: the whole GPU suite on the final commit, smoke, the fitted workload again (its definition now includes the
# reference's learning-rate schedule): snapshot after 400 iterations + the frame after 25 / 100 / 1600, the driver's bench command line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_check4; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -6 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
bash tools/snapshot.sh r06i_fitted fitted > $OUT/snap_fitted.log 2>&1
for it in 25 100 1600; do GSR_FIT_ITERS=$it timeout 900 python bench.py --workload fitted --no-next-rows --no-strict-parity --cpu-budget 6 2> $OUT/bench_fitted$it.err | tail -1 > $OUT/bench_fitted_$it.json; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
tail -n 2 $OUT/pytest.txt; tail -n 2 $OUT/smoke.txt
