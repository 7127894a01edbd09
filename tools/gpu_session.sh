#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_s26; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
# the driver's own command line, three times (how much a K = 20 region moves), then with the steady-state warm-up off
for r in 1 2 3; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-strict-parity 2>>$OUT/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver-style', d['value'], d['ms_per_step'], d['ms_per_step_spread']['blocks_ms'], d['warmup_extra_steps'], d['view_cache']['ms_per_step_without'])" | tee -a $OUT/driver_style.txt; done
for r in 1 2 3; do GSR_BENCH_NO_BRACKET=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-strict-parity 2>>$OUT/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver-style, no event brackets in the region', d['value'], d['ms_per_step'], d['ms_per_step_spread']['blocks_ms'], d['warmup_extra_steps'], d['view_cache']['ms_per_step_without'])" | tee -a $OUT/driver_style.txt; done
for r in 1 2 3; do GSR_BENCH_WARM_MS=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-strict-parity 2>>$OUT/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver-style, no extra warm-up', d['value'], d['ms_per_step'], d['ms_per_step_spread']['blocks_ms'], d['warmup_extra_steps'], d['view_cache']['ms_per_step_without'])" | tee -a $OUT/driver_style.txt; done
for wl in config2 config3 config4; do bash tools/snapshot.sh r05_$wl $wl > $OUT/snap_$wl.log 2>&1; done
for wl in init_state surfaces; do timeout 900 python bench.py --workload $wl 2> $OUT/bench_$wl.err | tail -1 > $OUT/bench_$wl.json; done
cp profiles/pmc_latest.json $OUT/pmc_latest.json
tail -3 $OUT/smoke.txt
