#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6: PACKED forward launch (as many waves as slots, tasks dealt boustrophedon in the view-cache order): A/B against the same source with packing off; parity suites
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/fwd_pack; mkdir -p $OUT; rm -f $OUT/ab.txt
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d.get('view_cache',{}).get('ms_per_step_without'), {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})" "$1" "$2"; }
run() { local wl=$1 name=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 --workload $wl 2>>$OUT/err.log | tail -1 | row $wl $name | tee -a $OUT/ab.txt; }
L() { echo "GSR_LIB=$PWD/gscream_amd/libgsraster_$1.so"; }
for wl in config2 config3 surfaces config4 init_state; do
  run $wl nopack $(L nopack); run $wl pack A=1; run $wl nopack $(L nopack); run $wl pack A=1
done
for it in 25 400 1600; do run fitted nopack_$it $(L nopack) GSR_FIT_ITERS=$it; run fitted pack_$it GSR_FIT_ITERS=$it; done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_precise.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -4 | tee $OUT/pytest.txt
