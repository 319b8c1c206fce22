#!/bin/bash
# usage: run.sh name...   -> bench each variants/libsage2ov_<name>.so (no CPU baseline), one line each
for v in "$@"; do
  SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-noisy-variant 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['phases_ms'].items()}, 'kern', round(r['roofline']['kernel_ms'],2), r['config']['edges_crc32'])
" >> gpurun_out/variants.log || echo "$v failed" >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
