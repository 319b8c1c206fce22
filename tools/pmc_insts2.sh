#!/bin/bash
# usage: tools/pmc_insts2.sh variant...   -> VALU / SALU / LDS / VMEM wave-instructions per read of the probe kernel, per variant library (tests/diag/variants/build), 10 M reads
export TMPDIR=/tmp
R=$PWD
for v in "$@"; do
  rm -rf $R/gpurun_out/pi_$v
  SAGE2OV_LIB=$R/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pi_$v -o x -- python3 tests/diag/probe_only.py 10000000 1 > $R/gpurun_out/pi_$v.log 2>&1
  python3 tools/pmc_sum.py $R/gpurun_out/pi_$v 8502556 | awk -v t=$v '{print t, $0}'
  rm -rf $R/gpurun_out/pi_$v
done
