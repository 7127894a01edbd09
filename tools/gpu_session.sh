#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s8; mkdir -p "$OUT"
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > "$OUT/pytest_all.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1
timeout 900 python bench.py 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_config2.json"
timeout 900 python bench.py --workload config5 --steps 50 --warmup 10 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_config5_1gpu.json"
timeout 900 python bench.py --gpus 2 --oversubscribe --backend gloo --no-cpu-baseline --no-next-rows 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_2ranks_testmode.json"
cat "$OUT/pytest_all.txt"; tail -3 "$OUT/smoke.txt"
python - <<'EOF'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5_s8/"
d = json.load(open(o + "bench_config2.json"))
print("config2", d["value"], d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("frac", "avg_launch_ms", "traffic")}, "valu", d["roofline"].get("valu"))
c = json.load(open(o + "bench_config5_1gpu.json"))
print("config5", c["value"], c["wall_clock"], c["per_rank"])
t = json.load(open(o + "bench_2ranks_testmode.json"))
print("2 ranks", t["value"], t["per_rank"])
EOF
