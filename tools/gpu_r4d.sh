#!/bin/bash
# round 4, step 4: band flags in the point list (backward checks flagged instances only), forward at 6 waves again; clamp-rare variant
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p "$OUT"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for WL in config2 config3 config4; do
  run cur A=1
  run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
  run clamprare GSR_LIB=$PWD/gscream_amd/libgsraster_clamprare.so
  run cur A=1
  run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
  run clamprare GSR_LIB=$PWD/gscream_amd/libgsraster_clamprare.so
done
GSR_LIB=$PWD/gscream_amd/libgsraster_count.so timeout 300 python tools/blend_counts.py config2 > "$OUT/counts.txt" 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^\[Gloo\]" > "$OUT/pytest_full.txt"; tail -15 "$OUT/pytest_full.txt" > "$OUT/pytest.txt"; grep -o "\[threshold flips\].*" "$OUT/pytest_full.txt" | sort | uniq -c | sort -rn | head -60 > "$OUT/flips.txt"
timeout 900 python bench.py 2>>"$OUT/err.log" | tail -1 > "$OUT/bench.json"
GSR_LIB=$PWD/gscream_amd/libgsraster_inf6.so timeout 600 python bench.py --no-cpu-baseline --no-strict-parity 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_inf6.json"
