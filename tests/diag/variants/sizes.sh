#!/bin/bash
# kernel time per read at different dataset sizes (same coverage)
for n in 500000 1000000 2000000 5000000 10000000; do
  g=$((n*3))
  timeout -k 10 200 python3 bench.py --reads $n --genome $g --steps 3 --warmup 1 --no-cpu-baseline --no-noisy-variant 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); u = r['config']['unique_reads']; k = r['roofline']['kernel_ms']
        print($n, 'unique', u, 'kern_ms', round(k,3), 'ns/read', round(k*1e6/u,3), 'index_ms', round(r['phases_ms']['index_ms'],2), 'step', round(r['ms_per_step'],2))
" >> gpurun_out/sizes.log
done
cat gpurun_out/sizes.log
