#!/bin/bash
# BASELINE.json configs[1]: 10 M x 150 bp, k=40 on one MI355X under rocprofv3 (kernel trace, then PMC passes, one derived counter per pass:
# FETCH_SIZE and WRITE_SIZE together exceed what one pass can collect).  Run from the repo root on a GPU box.
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/c2; mkdir -p $O
ARGS="--reads 10000000 --no-cpu-baseline --no-noisy-variant --no-step4 --no-scaling-model --steps 5 --warmup 1"     # configs[1] (bench.py defaults to configs[2])
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 bench.py $ARGS > $O/bench_stats.log 2>&1
echo stats done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_a -o c2 -- python3 bench.py $ARGS > $O/bench_pmc_a.log 2>&1
echo pmc a done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_b -o c2 -- python3 bench.py $ARGS > $O/bench_pmc_b.log 2>&1
echo pmc b done
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_c -o c2 -- python3 bench.py $ARGS > $O/bench_pmc_c.log 2>&1
echo pmc c done
grep '^{' $O/bench_stats.log | cut -c1-300
