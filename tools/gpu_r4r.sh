#!/bin/bash
# wave-level rank sort of medium buckets: parity suite + sort times on config 2 / 4 / the large-splat row
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p "$OUT"; : > "$OUT/ab.txt"
cat > /tmp/row.py <<'P'
import json,sys
d=json.loads(sys.stdin.read()); ti=d['next_rows']['train_iteration']
print(sys.argv[1], 'train_iteration', ti['ms_per_iteration'], 'kernels', ti['gpu_kernel_ms_sum'], 'R', ti.get('num_rendered'),
      {k.replace('void ','')[:28]:v for k,v in list(ti['gpu_top_kernels_us'].items())[:12]})
P
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_knn.py -x -q -m gpu 2>&1 | tail -3 | tee "$OUT/pytest.txt"
for rep in 1 2; do
  for WL in config2 config4; do run cur A=1; run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1; done
  for occ in -1 0; do
    timeout 600 python bench.py --no-cpu-baseline --no-strict-parity --occlusion $occ 2>>"$OUT/err.log" | tail -1 | python /tmp/row.py "occ=$occ" | tee -a "$OUT/ab.txt"
  done
done
