set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s40; mkdir -p $OUT
timeout 900 python bench.py --workload config5 --steps 20 --warmup 5 2> $OUT/c5.err | tail -1 > $OUT/config5_1gpu.json
timeout 900 python bench.py --gpus 2 --oversubscribe --backend gloo --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-strict-parity 2> $OUT/two.err | tail -1 > $OUT/two_ranks.json
for wl in init_state surfaces; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -- python bench.py --workload $wl --no-cpu-baseline --no-next-rows --no-strict-parity 2> $OUT/kt_$wl.err | tail -1 > $OUT/prof_$wl.json
  DB=$(find $OUT/kt_$wl -name "*_results.db" | head -1); python tools/rocprof_summary.py "$DB" > $OUT/${wl}_kernel_stats.md 2>>$OUT/err.log; rm -rf $OUT/kt_$wl
done
ls -la $OUT
