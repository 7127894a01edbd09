#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -6
timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-strict-parity --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d.get('stages',{}).items()})"
