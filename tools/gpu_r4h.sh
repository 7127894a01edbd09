#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} "$@" 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
WL=config4 run auto
WL=config4 run b1 --scatter-bands 1
WL=config4 run b4 --scatter-bands 4
WL=config2 run auto
WL=config3 run auto
for b in 0 1 2 4; do
  timeout 600 python bench.py --no-cpu-baseline --no-strict-parity --scatter-bands $b 2>>"$OUT/err.log" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); ti=d['next_rows']['train_iteration']; print('train_iteration bands=$b', ti['ms_per_iteration'], {k.replace('void ','')[:28]:v for k,v in list(ti['gpu_top_kernels_us'].items())[:9]})" | tee -a "$OUT/ab.txt"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 > "$OUT/pytest.txt"
