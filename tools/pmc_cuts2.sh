#!/bin/bash
# usage: tools/pmc_cuts2.sh variant...  -> per variant (tests/diag/variants/build/libsage2ov_<v>.so, e.g. the -DSAGE2OV_CUT=N builds): VALU / SALU / LDS / VMEM wave-instructions
# per read of the probe kernel (all its launches) and the kernel time -- differences of consecutive cuts are a stage's instructions
export TMPDIR=/tmp
R=$PWD
for v in "$@"; do
  rm -rf $R/gpurun_out/pc_$v
  SAGE2OV_LIB=$R/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pc_$v -o x -- python3 tests/diag/probe_only.py 10000000 1 > $R/gpurun_out/pc_$v.log 2>&1
  python3 - $R/gpurun_out/pc_$v $v <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_probe_fast" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
N = 8502556.0
print(sys.argv[2], " ".join(f"{k[9:]} {v / N:8.1f}" for k, v in sorted(acc.items())), end=" | ")
P
  grep "probe kernel" $R/gpurun_out/pc_$v.log | tail -1
  rm -rf $R/gpurun_out/pc_$v
done
