#!/bin/bash
# Round-end check on the GPU box: build check, smoke(), full -m gpu suite, default bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\[Gloo\]" | tail -6
timeout 900 python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/final/bench.json")))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["valu"] and d["roofline"]["valu"]["frac_of_issue_ceiling"], d["whole_iteration"]["frac_of_hbm_peak"])
print({k: round(v["avg_ms"] * 1e3, 1) for k, v in d["stages"].items()})
print({k: (v.get("ms") or v.get("forward_backward_ms") or v.get("ms_per_iteration")) for k, v in d["next_rows"].items()})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_torch_naive"]["value"])
PY
