#!/bin/bash
# usage (GPU box): bash tools/gpu_env_ab.sh "NAME=VALUE[,NAME=VALUE]" ...   -- bench the shipped library under each environment
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/envab; mkdir -p "$OUT"
WL=${WORKLOAD:-config2}
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
run base A=1
for v in "$@"; do run "$v" $(echo "$v" | tr ',' ' '); done
