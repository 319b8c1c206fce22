#!/bin/bash
# index build variants: index_ms at configs[2] (3 steps)
for spec in "ixw256:6" "ixw512:6" "ixw512:4" "ixw512:8" "ixw1024:4" "ixw1024:2" "ixw256:6"; do
  v=${spec%%:*}; g=${spec##*:}
  SAGE2OV_IXW_GRID_PER_CU=$g SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-noisy-variant --no-c2 --no-step4 --no-scaling-model 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$spec', round(r['ms_per_step'],2), 'index', round(r['phases_ms']['index_ms'],2), r['config']['edges_crc32'])
"
done
