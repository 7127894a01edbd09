#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload fitted --no-cpu-baseline --no-next-rows --no-strict-parity 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('fitted_run'))"
