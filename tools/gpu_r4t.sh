#!/bin/bash
# two-level counting sort only: full GPU suite + config 2/3/4 + next rows
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee "$OUT/pytest.txt"
for rep in 1 2; do
  for WL in config2 config3 config4; do run cur A=1; done
done
timeout 900 python bench.py --no-cpu-baseline > "$OUT/bench_full.json" 2>>"$OUT/err.log"
python - <<'P' | tee -a "$OUT/ab.txt"
import json
d=json.loads(open("/root/repo/gpurun_out/r4t/bench_full.json").read().strip().splitlines()[-1])
nr=d['next_rows']
print('value', d['value'], 'knn', nr['simple_knn']['ms'], 'train', nr['train_iteration']['ms_per_iteration'], nr['train_iteration']['gpu_kernel_ms_sum'], 'pipeline', nr['pipeline_decode_raster_loss']['ms_per_iteration'], 'fps', nr['render_fps']['rasterizer_bench_scene']['fps'], nr['render_fps']['standin_model_view']['fps'])
print(nr['render_fps']['rasterizer_bench_scene']['stages_us'])
P
