#!/bin/bash
# same-box A/B of the current library against the round-3 library (gscream_amd/libgsraster_r3.so) + optional variants ($@)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_r3; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for rep in 1 2 3; do
  for WL in config2 config4; do
    run cur A=1
    run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
    for v in "$@"; do run $v GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so; done
  done
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_precise.py tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -4 > "$OUT/pytest.txt"
