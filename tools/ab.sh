#!/bin/bash
# usage: tools/ab.sh VARIANT...   -- bench each gscream_amd/libgsraster_<VARIANT>.so (diagnostic A/B builds)
for v in "$@"; do
  GSR_LIB=$PWD/gscream_amd/libgsraster_$v.so timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], {k:v['avg_ms'] for k,v in d['stages'].items()})"
done
