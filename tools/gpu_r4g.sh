#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} "$@" 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for WL in config4; do
  run auto
  run b1 --scatter-bands 1
  run b2 --scatter-bands 2
  run b4 --scatter-bands 4
  run b8 --scatter-bands 8
  run b16 --scatter-bands 16
done
for WL in config2; do
  run auto
  run b2 --scatter-bands 2
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 > "$OUT/pytest.txt"
