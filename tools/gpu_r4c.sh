#!/bin/bash
# round 4, step 3: A/B current vs round 3 vs forward deep-first order experiment; scatter no-store floor at config 4; GPU suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p "$OUT"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for WL in config2 config4; do
  run cur A=1
  run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
  run order GSR_LIB=$PWD/gscream_amd/libgsraster_order.so
  run cur A=1
  run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
  run order GSR_LIB=$PWD/gscream_amd/libgsraster_order.so
  run nostore GSR_LIB=$PWD/gscream_amd/libgsraster_nostore.so
done
WL=config3 run cur A=1
WL=config3 run order GSR_LIB=$PWD/gscream_amd/libgsraster_order.so
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^\[Gloo\]" > "$OUT/pytest_full.txt"; tail -15 "$OUT/pytest_full.txt" > "$OUT/pytest.txt"; grep "threshold flips" "$OUT/pytest_full.txt" | sort | uniq -c | sort -rn | head -60 > "$OUT/flips.txt"
