#!/bin/bash
# occupancy variants of the clean-data probe kernel at configs[2]: "<variant>:<SAGE2OV_PROBE_SEQ>" ...
for spec in "$@"; do
  v=${spec%%:*}; q=${spec##*:}
  SAGE2OV_PROBE_SEQ=$q SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-noisy-variant --no-c2 --no-step4 --no-scaling-model 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$spec', round(r['ms_per_step'],2), 'probe', round(r['phases_ms']['probe_ms'],2), r['config']['edges_crc32'])
"
done
