#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_reference_vectors2.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -12
python tools/decode_probe.py 2>&1 | tail -1
