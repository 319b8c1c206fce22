#!/bin/bash
# usage: run3.sh name...  -> configs[2] headline (no side workloads) with variants/libsage2ov_<name>.so, one line each
for v in "$@"; do
  SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-noisy-variant --no-c2 --no-step4 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['phases_ms'].items()}, 'kern', round(r['roofline']['kernel_ms'],2), r['config']['edges_crc32'])
" >> gpurun_out/variants3.log || echo "$v failed" >> gpurun_out/variants3.log
done
cat gpurun_out/variants3.log
