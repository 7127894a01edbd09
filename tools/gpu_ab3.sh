#!/bin/bash
# usage (GPU box): bash tools/gpu_ab3.sh VARIANT...   -- bench the shipped library, then each gscream_amd/libgsraster_<VARIANT>.so
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p "$OUT"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
run base A=1
for v in "$@"; do run $v GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so; done
