#!/bin/bash
# instruction counts of the probe kernel per stage: PMC pass over builds that drop every read after stage N (variants/libsage2ov_cutN.so, -DSAGE2OV_CUT=N)
export TMPDIR=/tmp
R=$PWD
run() { tag=$1; lib=$2; SAGE2OV_LIB=$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pc_$tag -o x -- python3 tests/diag/probe_only.py 10000000 3 > $R/gpurun_out/pc_$tag.log 2>&1
  python3 tools/pmc_sum.py $R/gpurun_out/pc_$tag 8502556 | grep probe_fast | awk -v t=$tag '{printf "%s %s %s | ", t, $(NF-7), $NF} END {print ""}'; grep "probe kernel" $R/gpurun_out/pc_$tag.log | tail -1; }
for c in "$@"; do
  if [ "$c" = full ]; then run full $R/sage2_amd/libsage2ov.so; else run cut$c $R/variants/libsage2ov_cut$c.so; fi
done
