#!/bin/bash
# Round 6: the rotation block (64 shuffled views): final library vs the first session's (40601fb) on one box -- why is the cached variant's
# wall time above its kernel sum now?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/rotation_ab; mkdir -p $OUT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['rotation']
print(sys.argv[1], d['ms_per_step'], '| rot with', r['ms_per_step'], round(sum(r['stage_us'].values()),1), '| without', r['ms_per_step_no_view_cache'], round(sum(r['stage_us_no_view_cache'].values()),1), '| static', r['ms_per_step_static_gaussians'], round(sum(r['stage_us_static_gaussians'].values()),1))" "$1"; }
for k in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity 2>>$OUT/err.log | tail -1 | show final
GSR_LIB=$PWD/gscream_amd/libgsraster_s1.so GSR_SKIP_ABI_CHECK=1 GSR_SEG2=8 GSR_T2_LEN=1 GSR_T2_N=0 GSR_SEG3_LEN=1 timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity 2>>$OUT/err.log | tail -1 | show session1
done
