#!/bin/bash
# A/B of library variants on the GPU box: quick parity subset + bench stage times, two rounds interleaved.
# usage: bash tools/gpu_ab.sh [-w workload] name1 name2 ...   ("base" = the shipped libgsraster.so, others = libgsraster_<name>.so)
set -u
WL=config2
if [ "${1:-}" = "-w" ]; then WL=$2; shift 2; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p "$OUT"
lib() { if [ "$1" = base ]; then echo "$GRAFT_REPO_ROOT/gscream_amd/libgsraster.so"; else echo "$GRAFT_REPO_ROOT/gscream_amd/libgsraster_$1.so"; fi; }
for n in "$@"; do
  if [ "$n" != base ]; then
    GSR_LIB=$(lib $n) timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_forward_backward or slab_scene or config1 or determinism or more_cases" 2>&1 | tail -2 | sed "s/^/[$n parity] /"
  fi
done
for round in 1 2; do
  for n in "$@"; do
    GSR_LIB=$(lib $n) timeout 600 python bench.py --workload $WL --steps 100 --warmup 20 --no-cpu-baseline --no-next-rows --no-strict-parity 2> "$OUT/$n.err" | tail -1 > "$OUT/$n.json"
    python - "$n" "$OUT/$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"[{sys.argv[1]:>8s}] {d['value']:8.1f} it/s {d['ms_per_step']:.4f} ms | " + " ".join(f"{k.split('_')[0][:4]}{k.split('_')[-1][:3]}={v['avg_ms'] * 1e3:.1f}" for k, v in d["stages"].items()))
PY
  done
done
