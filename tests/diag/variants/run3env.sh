#!/bin/bash
# usage: run3env.sh "ENV=1 ..." ... -> configs[2] headline with the default library under each environment
for e in "$@"; do
  env $e timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-noisy-variant --no-c2 --no-step4 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$e', round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['phases_ms'].items()}, 'kern', round(r['roofline']['kernel_ms'],2), r['config']['edges_crc32'])
"
done
