#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_s3; mkdir -p "$OUT"
GSR_LIB=$PWD/gscream_amd/libgsraster_tband9.so GSR_SKIP_ABI_CHECK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > "$OUT/pytest_tband9.txt"
timeout 2400 python -m pytest tests/test_render_call.py tests/test_gpu_parity.py tests/test_gpu_precise.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -15 > "$OUT/pytest_shipped.txt"
TAG=r5_s3 WORKLOADS="config2 config4" REPEAT=2 bash tools/gpu_ab.sh notreplay tband0 tband3 > /dev/null 2>&1
for cfg in "1 1000000 1008 567 1 0 0" "2 1000000 1008 567 1 1 1" "3 2000000 1920 1080 1 1 1"; do
  timeout 900 python tools/full_size_oracle_check.py $cfg 2>>"$OUT/err.log" | tail -1 > "$OUT/fullsize_$(echo $cfg | cut -c1).json"
done
# init_state parity: shipped vs a 1e-2 wide alpha guard band (are its outliers alpha flips the 3e-5 band misses?) vs the parity build
for v in "" band2 precise; do
  GSR_LIB=$PWD/gscream_amd/libgsraster${v:+_$v}.so GSR_SKIP_ABI_CHECK=1 timeout 900 python bench.py --workload init_state --no-strict-parity --cpu-budget 2 2>>"$OUT/err.log" | tail -1 > "$OUT/bench_init_state_${v:-shipped}.json"
done
cat "$OUT/pytest_tband9.txt" "$OUT/pytest_shipped.txt" "$OUT/ab.txt"
python - <<'EOF'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5_s3/"
for c in "123":
    try:
        pc = json.load(open(o + f"fullsize_{c}.json"))["parity_check"]
        env = pc.get("order_noise_envelope") or {}
        print("cfg", c, "px", pc["px_gt_1e-4"], "grad", pc["grad_elems_gt_1e-3"], pc["grad_elems_by_cause"], "stops", pc.get("last_contributor_differs"),
              "envelope in/out", env.get("elements_inside"), env.get("elements_outside"), "Trel", pc.get("final_T_max_rel_where_same_stop"))
    except Exception as ex:
        print("cfg", c, "error", ex)
for v in ("shipped", "band2", "precise"):
    try:
        d = json.load(open(o + f"bench_init_state_{v}.json"))
        pc = d["parity_check"]
        print("init_state", v, d["value"], {k: round(x["avg_ms"] * 1e3, 1) for k, x in d["stages"].items()})
        print("   parity", {k: pc.get(k) for k in ("px_gt_1e-4", "max_abs", "grad_elems_gt_1e-3", "worst_rel", "grad_elems_by_cause", "last_contributor_differs", "per_family")},
              "env in/out", (pc.get("order_noise_envelope") or {}).get("elements_inside"), (pc.get("order_noise_envelope") or {}).get("elements_outside"))
        for k in ("train_iteration", "train_iteration_init_state"):
            r = d.get("next_rows", {}).get(k)
            if r:
                print("   ", k, {kk: r.get(kk) for kk in ("ms_per_iteration", "host_ms", "gpu_kernel_ms_sum", "num_rendered", "error")}, dict(list((r.get("gpu_top_kernels_us") or {}).items())[:6]))
    except Exception as ex:
        print("init_state", v, "error", ex)
EOF
