#!/bin/bash
# two-level bucket sort: register budget of the 256-thread sort kernels (6 / 7 / 8 waves per SIMD)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4s; mkdir -p "$OUT"; : > "$OUT/ab.txt"
cat > /tmp/row.py <<'P'
import json,sys
d=json.loads(sys.stdin.read()); ti=d['next_rows']['train_iteration']
print(sys.argv[1], 'train_iteration', ti['ms_per_iteration'], 'kernels', ti['gpu_kernel_ms_sum'], 'R', ti.get('num_rendered'),
      {k.replace('void ','')[:28]:v for k,v in list(ti['gpu_top_kernels_us'].items())[:12] if 'sort' in k})
P
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for rep in 1 2; do
  for v in "" two8 two7; do
    L=$PWD/gscream_amd/libgsraster${v:+_$v}.so
    for WL in config2 config4; do run "${v:-both6}" GSR_LIB=$L; done
    GSR_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-strict-parity 2>>"$OUT/err.log" | tail -1 | python /tmp/row.py "${v:-both6}" | tee -a "$OUT/ab.txt"
  done
done
