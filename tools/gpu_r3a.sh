#!/bin/bash
# Round 3, first measurement pass: full -m gpu suite, host profile (step time vs P with per-stage GPU sums), default bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^\[Gloo\]" > "$OUT/pytest.txt"; tail -15 "$OUT/pytest.txt"
grep -E "^\[(shipped|precise|mask flips)\]" "$OUT/pytest.txt" | head -40
timeout 600 python tools/host_profile.py > "$OUT/host_profile.txt" 2>&1; head -30 "$OUT/host_profile.txt"
timeout 1200 python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"; tail -3 "$OUT/bench.err"
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3a/bench.json")))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("host_ms_per_step"), d["roofline"]["traffic_source"]["status"])
print({k: round(v["avg_ms"] * 1e3, 1) for k, v in d["stages"].items()})
print(json.dumps(d.get("strict_parity_build"))[:600])
for k, v in d["next_rows"].items():
    print(k, json.dumps(v)[:900])
PY
