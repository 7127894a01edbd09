#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s2; mkdir -p "$OUT"
# 1. the ripple replay on every pixel that stops (GSR_TBAND = 0.9): the whole parity suite through it
GSR_LIB=$PWD/gscream_amd/libgsraster_tband9.so GSR_SKIP_ABI_CHECK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > "$OUT/pytest_tband9.txt"
# 2. new tests of the round + the parity suite on the shipped library
timeout 1500 python -m pytest tests/test_render_call.py tests/test_gpu_parity.py tests/test_gpu_precise.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -15 > "$OUT/pytest_shipped.txt"
# 3. what the replay costs now: code present but never taken (band 0), shipped (1e-4), 7 waves per SIMD with two spilled registers, band 1e-3
TAG=r5_s2 WORKLOADS="config2 config4" REPEAT=2 bash tools/gpu_ab.sh notreplay tband0 fw7 tband3 > /dev/null 2>&1
# 4. full-size element-wise check of the shipped library (config 2 / 3 / 4 + surfaces is not a slab: skipped here)
for cfg in "1 1000000 1008 567 1 0 0" "2 1000000 1008 567 1 1 1" "3 2000000 1920 1080 1 1 1"; do
  timeout 900 python tools/full_size_oracle_check.py $cfg 2>>"$OUT/err.log" | tail -1 > "$OUT/fullsize_$(echo $cfg | cut -c1).json"
done
# 5. the init-state workload: bench line + train_iteration rows
timeout 900 python bench.py --workload init_state --no-strict-parity --cpu-budget 4 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_init_state.json"
cat "$OUT/pytest_tband9.txt" "$OUT/pytest_shipped.txt" "$OUT/ab.txt"
python - <<'EOF'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5_s2/"
for c in "123":
    try:
        pc = json.load(open(o + f"fullsize_{c}.json"))["parity_check"]
        env = pc.get("order_noise_envelope") or {}
        print("cfg", c, "px", pc["px_gt_1e-4"], "grad", pc["grad_elems_gt_1e-3"], pc["grad_elems_by_cause"], "stops", pc.get("last_contributor_differs"),
              "envelope in/out", env.get("elements_inside"), env.get("elements_outside"))
        for e in env.get("elements", []):
            print("   ", e["family"], e["component"], e["cause"], "ours", e["ours"], "dbl", e["oracle_double"], "ref32 [", e["reference_fp32_orders_min"], e["reference_fp32_orders_max"], "] ratio", round(e["ours_minus_double_over_envelope_halfwidth"], 2), e["inside_envelope"])
    except Exception as ex:
        print("cfg", c, "error", ex)
try:
    d = json.load(open(o + "bench_init_state.json"))
    print("init_state", d["value"], d["ms_per_step"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["stages"].items()})
    print(json.dumps(d["scene_stats"]))
    print("parity", {k: d["parity_check"].get(k) for k in ("px_gt_1e-4", "grad_elems_gt_1e-3", "grad_elems_by_cause", "last_contributor_differs")})
    for k in ("train_iteration", "train_iteration_init_state"):
        r = d["next_rows"][k]
        print(k, {kk: r.get(kk) for kk in ("ms_per_iteration", "host_ms", "gpu_kernel_ms_sum", "num_rendered", "num_occluded", "error")}, r.get("what", "")[:200])
        print("   ", r.get("gpu_top_kernels_us"))
except Exception as ex:
    print("init_state error", ex)
EOF
