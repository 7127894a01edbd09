#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6: per-Gaussian backward, compact path (zero rows filled by the backward blend on the side) -- test + A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/compact_ab; mkdir -p $OUT; rm -f $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_gauss_bwd_compact.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > $OUT/pytest.txt
tail -n 5 $OUT/pytest.txt
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})" "$1" "$2"; }
for wl in config2 config3 config4 surfaces init_state; do for mode in 0 1 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 --workload $wl --gauss-bwd-compact $mode 2>>$OUT/err.log | tail -1 | row $wl compact=$mode | tee -a $OUT/ab.txt
done; done
