#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6: the `fitted` workload (gscream_amd/fit.py: a short optimisation run through the HIP rows) -- its tests, then the bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/fitted; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fit.py -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -25 > $OUT/pytest.txt
tail -n 12 $OUT/pytest.txt
timeout 900 python bench.py --workload fitted --no-next-rows --no-strict-parity 2> $OUT/bench.err | tail -1 > $OUT/bench_fitted.json
tail -5 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/fitted/bench_fitted.json"))
print(d["value"], d["ms_per_step"], d.get("fitted_run"))
print({k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})
print(d.get("scene_stats"))
print(d.get("parity_check"))
PY
