#!/bin/bash
# Round snapshot on the GPU box: PMC passes FIRST (they refresh profiles/pmc_latest.json, whose source hash the bench line checks before
# it replays `roofline.traffic` / `valu`), then the official bench line, then rocprofv3 kernel stats of the same command.
# usage (repo root, on the GPU box): bash tools/snapshot.sh <tag> [workload]      -> gpurun_out/snap_<tag>/*
set -u
TAG=${1:-X}; WL=${2:-config2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/snap_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
bash tools/pmc_run.sh $WL "$OUT/pmc" > "$OUT/pmc.log" 2>&1
cp profiles/pmc_latest.json "$OUT/pmc_latest_before.json" 2>/dev/null
python tools/pmc_summary.py "$OUT/pmc" $WL > "$OUT/pmc.md" 2>> "$OUT/pmc.log"
cp profiles/pmc_latest.json "$OUT/pmc_latest.json"
rm -rf "$OUT/pmc"
timeout 900 python bench.py --workload $WL 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python bench.py --workload $WL --no-cpu-baseline --no-next-rows 2> "$OUT/kt.err" | tail -1 > "$OUT/bench_profiled.json"
DB=$(find "$OUT/kt" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" > "$OUT/kernel_stats.md" 2>> "$OUT/kt.err"
rm -rf "$OUT/kt"
ls -la "$OUT"
