#!/bin/bash
# scatter: first trip's loads in front of the prologue (A/B against the build without), then the full bench line (render_fps fix)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tee "$OUT/pytest.txt"
for rep in 1 2 3; do
  for WL in config2 config4; do
    run pre A=1
    run nopre GSR_LIB=$PWD/gscream_amd/libgsraster_nopre.so
  done
done
timeout 900 python bench.py > "$OUT/bench_full.json" 2>>"$OUT/err.log"
python - <<'P' | tee -a "$OUT/ab.txt"
import json
d=json.loads(open("/root/repo/gpurun_out/r4v/bench_full.json").read().strip().splitlines()[-1])
nr=d['next_rows']
print('value', d['value'], 'rgb', nr['rgb_loss']['ms'], 'depth', nr['depth_loss']['ms'], 'knn', nr['simple_knn']['ms'], 'train', nr['train_iteration']['ms_per_iteration'], nr['train_iteration'].get('ms_per_iteration_blocks'), nr['train_iteration']['gpu_kernel_ms_sum'], 'pipeline', nr['pipeline_decode_raster_loss']['ms_per_iteration'])
print('fps', nr['render_fps']['rasterizer_bench_scene']['fps'], nr['render_fps']['rasterizer_bench_scene'].get('ms_per_frame_blocks'), nr['render_fps']['standin_model_view']['fps'])
P
