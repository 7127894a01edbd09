#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_s24; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -40 > $OUT/pytest.txt
for wl in config2 config3 config4 surfaces init_state; do for r in 1 2; do for mode in "" "--no-view-cache"; do
  timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 --workload $wl $mode 2>>$OUT/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', '[$mode]', d['value'], d['ms_per_step'], d['ms_per_step_spread']['blocks_ms'], 'host', d['per_rank'][0]['host_ms'], {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})" | tee -a $OUT/view_cache.txt
done; done; done
tail -5 $OUT/pytest.txt
