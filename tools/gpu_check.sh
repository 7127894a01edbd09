#!/bin/bash
# usage (GPU box): bash tools/gpu_check.sh [pytest files...]  -- bench (config2/3/4 stage times) + the given -m gpu tests
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p "$OUT"
for wl in config2 config3 config4; do
  timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload $wl 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
done
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -12; fi
