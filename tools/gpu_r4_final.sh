#!/bin/bash
# round-4 evidence: full GPU suite, then bench line + rocprofv3 kernel stats + PMC passes for configs 2, 3 and 4
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4final; mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\[Gloo\]" > "$OUT/pytest_full.txt"; tail -6 "$OUT/pytest_full.txt" > "$OUT/pytest.txt"; grep -o "\[threshold flips\].*" "$OUT/pytest_full.txt" | sort | uniq -c > "$OUT/flips.txt"
for wl in config2 config3 config4; do
  bash tools/snapshot.sh r04_$wl $wl > "$OUT/snap_$wl.log" 2>&1
done
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1
