#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, as left at the end of the round: the full check -- GPU tests, smoke, the driver's bench command line, the tracked snapshots of
# every workload (PMC first, bench line, rocprofv3 kernel stats), the fitted frame along the optimisation run, both fuzz sweeps.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/full_check; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
for wl in config2 config3 config4 fitted init_state surfaces; do bash tools/snapshot.sh r06_$wl $wl > $OUT/snap_$wl.log 2>&1; done
for it in 25 100 1600; do GSR_FIT_ITERS=$it timeout 900 python bench.py --workload fitted --no-next-rows --no-strict-parity --cpu-budget 6 2> $OUT/bench_fitted$it.err | tail -1 > $OUT/bench_fitted_$it.json; done
FUZZ_KNOBS=1 timeout 1500 python tools/fuzz_parity.py 60 > $OUT/fuzz_60_seed1000.txt 2>&1
FUZZ_KNOBS=1 timeout 2400 python tools/fuzz_parity.py 120 2000 > $OUT/fuzz_120_seed2000.txt 2>&1
tail -n 3 $OUT/pytest.txt; tail -n 3 $OUT/smoke.txt; tail -n 2 $OUT/fuzz_60_seed1000.txt $OUT/fuzz_120_seed2000.txt
