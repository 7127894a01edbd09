#!/bin/bash
# blend kernels under other machine-scheduler strategies (the shipped build uses iterative-ilp)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items() if 'blend' in k})" | tee -a "$OUT/ab.txt"
}
for rep in 1 2; do
  for WL in config2 config3; do
    run cur A=1
    for v in max-ilp iterative-minreg iterative-maxocc max-memory-clause default; do [ -f gscream_amd/libgsraster_bs_$v.so ] && run $v GSR_LIB=$PWD/gscream_amd/libgsraster_bs_$v.so; done
  done
done
