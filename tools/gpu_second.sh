#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/second; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_issue tools/microbench/valu_issue.hip && timeout 300 /tmp/valu_issue > "$OUT/valu_issue.md" 2>&1
cat "$OUT/valu_issue.md"
timeout 1200 python -m pytest tests/test_reference_vectors2.py tests/test_gpu_precise.py tests/test_gpu_decode.py tests/test_gpu_parity.py -m gpu -q -s -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
grep -v "^\[Gloo\]" "$OUT/pytest.log" | tail -60
