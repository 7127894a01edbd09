#!/bin/bash
# The one A/B runner (replaces the per-experiment gpu_r*.sh scripts of rounds 1-4).  On the GPU box, from the repo root:
#   bash tools/gpu_ab.sh [variant ...]
# runs bench.py (stage times from HIP events) on the shipped library, then on every gscream_amd/libgsraster_<variant>.so
# (built here with `make -C gscream_amd/csrc variant SRC=... NAME=<variant> FLAGS=...`) through GSR_LIB, then on the shipped
# one again (box drift), for every workload in $WORKLOADS; $TESTS (pytest arguments) run afterwards.  Knobs (environment):
#   WORKLOADS="config2 config3 config4"   STEPS=50 WARMUP=10   REPEAT=1   TESTS="tests/test_gpu_parity.py"   TAG=ab
#   BENCH_ARGS="--occlusion 0"            extra bench.py arguments
# Output: gpurun_out/$TAG/ab.txt (one line per run), err.log.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-ab}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
WORKLOADS=${WORKLOADS:-config2}; STEPS=${STEPS:-50}; WARMUP=${WARMUP:-10}; REPEAT=${REPEAT:-1}
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})" "$1" "$2"; }
run() { local wl=$1 name=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps $STEPS --warmup $WARMUP --workload $wl ${BENCH_ARGS:-} 2>>"$OUT/err.log" | tail -1 | row $wl $name | tee -a "$OUT/ab.txt"
}
for wl in $WORKLOADS; do for r in $(seq $REPEAT); do
  run $wl shipped A=1
  for v in "$@"; do run $wl $v GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so GSR_SKIP_ABI_CHECK=1; done
done; [ $# -gt 0 ] && run $wl shipped_again A=1; done
if [ -n "${TESTS:-}" ]; then timeout 2400 python -m pytest $TESTS -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -15 | tee -a "$OUT/ab.txt"; fi
