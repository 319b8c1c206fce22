#!/bin/bash
# configs[1] with 0.1 % errors (bench.py's noisy_variant) under rocprofv3 --kernel-trace --stats: two whole steps of tests/diag/noisy_step.py.  Run from the repo root on a GPU box.
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/noisy; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o n -- python3 tests/diag/noisy_step.py 10000000 2 > $O/steps.log 2>&1
grep "^index" $O/steps.log
