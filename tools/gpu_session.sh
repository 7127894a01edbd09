#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6: new segment layout (7 x L, 6 x 3 L, 13 x 8 L) + deep-first flag: the whole GPU suite, the fuzz sweep with every knob, bench lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/layout_check; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > $OUT/pytest.txt
tail -n 3 $OUT/pytest.txt
FUZZ_KNOBS=1 timeout 1500 python tools/fuzz_parity.py 60 > $OUT/fuzz.txt 2>&1
tail -n 3 $OUT/fuzz.txt | cut -c1-300
row() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})" "$1" "$2"; }
for it in 25 100 400 1600; do GSR_FIT_ITERS=$it timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 --workload fitted 2>>$OUT/err.log | tail -1 | row fitted final_$it | tee -a $OUT/ab.txt; done
for wl in init_state config2 config3 config4 surfaces; do timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 --workload $wl 2>>$OUT/err.log | tail -1 | row $wl final | tee -a $OUT/ab.txt; done
