#!/bin/bash
# First GPU call of a round: full -m gpu suite, the VALU issue microbenchmark, one official bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/first; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_issue tools/microbench/valu_issue.hip && timeout 300 /tmp/valu_issue > "$OUT/valu_issue.md" 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout 900 python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/first/bench.json")))
print(d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["stages"].items()})
print({k:(v.get("ms") or v.get("forward_backward_ms") or v.get("ms_per_iteration")) for k,v in d.get("next_rows",{}).items()})
PY
head -80 "$OUT/valu_issue.md"
