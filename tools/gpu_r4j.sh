#!/bin/bash
# train-iteration row with the occlusion cut-off automatic / off, config 2 / 4 stage times
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p "$OUT"; : > "$OUT/ab.txt"
cat > /tmp/row.py <<'P'
import json,sys
d=json.loads(sys.stdin.read()); ti=d['next_rows']['train_iteration']
print(sys.argv[1], d['value'], 'train_iteration', ti['ms_per_iteration'], 'R', ti.get('num_rendered'), 'occluded', ti.get('num_occluded'),
      {k.replace('void ','')[:28]:v for k,v in list(ti['gpu_top_kernels_us'].items())[:12]})
print('   pipeline', d['next_rows']['pipeline_decode_raster_loss']['ms_per_iteration'], 'fps', d['next_rows']['render_fps']['standin_model_view']['fps'], d['next_rows']['render_fps']['rasterizer_bench_scene']['fps'])
P
for b in "--occlusion -1" "--occlusion 0" "--occlusion -1" "--occlusion 0"; do
  timeout 600 python bench.py --no-cpu-baseline --no-strict-parity $b 2>>"$OUT/err.log" | tail -1 | python /tmp/row.py "bench[$b]" | tee -a "$OUT/ab.txt"
done
run() { local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} "$@" 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
WL=config4 run auto
WL=config4 run b1 --scatter-bands 1
WL=config4 run occ --occlusion 1
WL=config2 run auto
WL=config2 run occ --occlusion 1
