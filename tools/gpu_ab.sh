#!/bin/bash
# A/B of blend variants + forward occupancy sweep; prints stage times per variant
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p "$OUT"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
if [ "${1:-}" = "micro" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_issue tools/microbench/valu_issue.hip && timeout 600 /tmp/valu_issue > "$OUT/valu_issue.md" 2>&1
  grep -E "\| 8 \|" "$OUT/valu_issue.md" | tail -8
fi
run base A=1
for v in sel w6 selw6; do run $v GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so; done
for pad in 3500 5000 7000 10000 13000; do run fwdpad$pad GSR_FWD_LDS_PAD=$pad; done
