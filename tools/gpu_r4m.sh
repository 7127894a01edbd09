#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for rep in 1 2; do
for WL in config4; do
  run cur A=1
  run heavy2k GSR_LIB=$PWD/gscream_amd/libgsraster_heavy2k.so
  run heavy4k GSR_LIB=$PWD/gscream_amd/libgsraster_heavy4k.so
done
done
