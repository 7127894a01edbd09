#!/bin/bash
# robustness probe: config 2's sizes on a surface-shaped cloud (clustered depths, early saturation)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], 'R', d['config']['num_rendered'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for WL in surfaces config2; do run cur A=1; run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1; done
timeout 900 python bench.py --workload surfaces --no-next-rows --no-strict-parity > "$OUT/bench_surfaces.json" 2>>"$OUT/err.log"
python - <<'P' | tee -a "$OUT/ab.txt"
import json
d=json.loads(open("/root/repo/gpurun_out/r4w/bench_surfaces.json").read().strip().splitlines()[-1])
print('surfaces', d['value'], d['ms_per_step'], 'parity', d.get('parity_check'), 'cpu', d['cpu_baseline']['value'])
P
python tools/sort_buckets_probe.py surfaces 2>&1 | grep -v Warn | tail -2 | tee -a "$OUT/ab.txt"
