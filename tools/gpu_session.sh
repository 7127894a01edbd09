#!/bin/bash
# The CURRENT GPU session's command list.  usage: gpurun -- 'bash tools/gpu_session.sh'
# Round 6, second session: third tier shipped -- the whole GPU suite, smoke, and the 60-case fuzz sweep (every knob bit-identical)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/tier3_check; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^\[Gloo\]" | tail -8 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 1500 FUZZ_KNOBS=1 python tools/fuzz_parity.py 60 > $OUT/fuzz.txt 2>&1
tail -n 3 $OUT/pytest.txt; tail -n 2 $OUT/smoke.txt; tail -n 6 $OUT/fuzz.txt
