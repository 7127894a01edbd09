#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s38; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > $OUT/pytest.txt
timeout 900 python bench.py --workload config2 2> $OUT/bench_config2.err | tail -1 > $OUT/bench_config2.json
for r in 1 2 3; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-strict-parity 2>>$OUT/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver-style', d['value'], d['ms_per_step'], d['ms_per_step_spread']['blocks_ms'], d['warmup_extra_steps'], d['view_cache']['ms_per_step_without'])" | tee -a $OUT/driver_style.txt; done
tail -3 $OUT/pytest.txt
