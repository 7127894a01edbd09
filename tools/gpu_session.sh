#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s5; mkdir -p "$OUT"
timeout 900 python tools/init_state_diag.py > "$OUT/diag.txt" 2>"$OUT/diag.err"
GSR_LIB=$PWD/gscream_amd/libgsraster_tband9.so GSR_SKIP_ABI_CHECK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > "$OUT/pytest_tband9.txt"
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_precise.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -15 > "$OUT/pytest_shipped.txt"
TAG=r5_s5 WORKLOADS="config2 config3 config4" REPEAT=2 bash tools/gpu_ab.sh notreplay > /dev/null 2>&1
for cfg in "1 1000000 1008 567 1 0 0" "2 1000000 1008 567 1 1 1" "3 2000000 1920 1080 1 1 1"; do
  timeout 900 python tools/full_size_oracle_check.py $cfg 2>>"$OUT/err.log" | tail -1 > "$OUT/fullsize_$(echo $cfg | cut -c1).json"
done
cat "$OUT/diag.txt"; tail -3 "$OUT/diag.err"; cat "$OUT/pytest_tband9.txt" "$OUT/pytest_shipped.txt" "$OUT/ab.txt"
python - <<'EOF'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5_s5/"
for c in "123":
    try:
        pc = json.load(open(o + f"fullsize_{c}.json"))["parity_check"]
        env = pc.get("order_noise_envelope") or {}
        print("cfg", c, "px", pc["px_gt_1e-4"], "grad", pc["grad_elems_gt_1e-3"], pc["grad_elems_by_cause"], "stops", pc.get("last_contributor_differs"),
              "envelope in/out", env.get("elements_inside"), env.get("elements_outside"))
    except Exception as ex:
        print("cfg", c, "error", ex)
EOF
