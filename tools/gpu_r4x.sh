#!/bin/bash
# -fno-slp-vectorize for the contract-off kernels (v_pk_mul / v_pk_add issue like two plain ops; the pairing costs v_mov)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p "$OUT"; : > "$OUT/ab.txt"
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-next-rows --steps 50 --warmup 10 --workload ${WL:-config2} 2>>"$OUT/err.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-config2} $name', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['stages'].items()})" | tee -a "$OUT/ab.txt"
}
for rep in 1 2; do
  for WL in config2 config4; do
    run cur A=1
    for v in ilp_gauss_bwd maxilp_gauss_bwd ilp_binning ilp_preprocess; do run $v GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so; done
  done
done
