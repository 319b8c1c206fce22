#!/bin/bash
export TMPDIR=/tmp
R=$PWD
run() { tag=$1; shift; timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pi_$tag -o x -- python3 tests/diag/phase_gaps.py 10000000 > $R/gpurun_out/pi_$tag.log 2>&1; python3 tools/pmc_sum.py $R/gpurun_out/pi_$tag 8502556 | awk -v t=$tag '{print t, $0}'; }
run new
SAGE2OV_LIB=$R/variants/libsage2ov_a.so run olda
