#!/bin/bash
# The CURRENT GPU session's command list (one file, rewritten per gpurun call; the parametrised pieces it calls --
# tools/gpu_ab.sh, tools/snapshot.sh, tools/pmc_run.sh -- are the reusable ones).  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, second session: the full check -- GPU tests, smoke, the driver's bench command line (with the fit_run row), the `fitted` snapshot
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/full_check2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_driver_style.json
bash tools/snapshot.sh r06_fitted fitted > $OUT/snap_fitted.log 2>&1
tail -n 3 $OUT/pytest.txt; tail -n 3 $OUT/smoke.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/full_check2/bench_driver_style.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(d["next_rows"].get("fit_run"))
print({k:(v.get("ms_per_iteration"), v.get("host_ms"), v.get("gpu_kernel_ms_sum")) for k,v in d["next_rows"].items() if isinstance(v,dict) and "ms_per_iteration" in v})
PY
