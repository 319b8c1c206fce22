#!/bin/bash
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-noisy-variant --no-step4 --steps 3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['ms_per_step'],2), round(r['phases_ms']['index_ms'],2), round(r['roofline']['kernel_ms'],2))
"; done
