#!/bin/bash
# round 4, step 2: same-box A/B of the round-3 library against the current one, GPU suite, bench with parity_check + render_fps
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p "$OUT"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for WL in config2 config4; do
  run cur A=1
  run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
  run cur A=1
  run r3 GSR_LIB=$PWD/gscream_amd/libgsraster_r3.so GSR_SKIP_ABI_CHECK=1
done
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > "$OUT/pytest.txt"
timeout 900 python bench.py 2>>"$OUT/err.log" | tail -1 > "$OUT/bench.json"
