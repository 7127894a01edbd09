#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6: the first tier's last 3 / 4 segments cut in two (backward tasks of 32 instances at the end of the grid: a shorter drain?) -- A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/t1_half; mkdir -p $OUT; rm -f $OUT/ab.txt
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})" "$1" "$2"; }
run() { local wl=$1 name=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 --workload $wl 2>>$OUT/err.log | tail -1 | row $wl $name | tee -a $OUT/ab.txt; }
H3="GSR_LIB=$PWD/gscream_amd/libgsraster_h3.so GSR_SKIP_ABI_CHECK=1 GSR_T1_HALF=3"
H4="GSR_LIB=$PWD/gscream_amd/libgsraster_h4.so GSR_SKIP_ABI_CHECK=1 GSR_T1_HALF=4"
for wl in config2 config3 surfaces config4 init_state; do
  run $wl shipped A=1; run $wl half3 $H3; run $wl half4 $H4; run $wl shipped A=1; run $wl half3 $H3
done
