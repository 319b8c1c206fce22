#!/bin/bash
# usage: [READS=10000000] tools/pmc_lines.sh variant...  -> memory-side (fabric) read requests of the probe kernel per read, per variant library: TCC_EA0_RDREQ_sum (every one
# a 128-byte line on gfx950: MI355X_MICROARCH.md, HBM section) -- lines per read = requests / unique reads
export TMPDIR=/tmp
R=$PWD; N=${READS:-10000000}
for v in "$@"; do
  rm -rf $R/gpurun_out/pl_$v
  SAGE2OV_LIB=$R/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pl_$v -o x -- python3 tests/diag/probe_only.py $N 1 > $R/gpurun_out/pl_$v.log 2>&1
  python3 - $R/gpurun_out/pl_$v $v $R/gpurun_out/pl_$v.log <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "probe" if "k_probe_fast" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0][-28:]
        acc[k] += float(r["Counter_Value"])
n = float(re.findall(r"unique reads (\d+)", open(sys.argv[3]).read())[-1])
print(sys.argv[2], f"probe kernel: {acc['probe'] / n:7.2f} lines/read ({acc['probe'] * 128 / 1e9:.1f} GB);", " ".join(f"{k} {v * 128 / 1e9:.2f}GB" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:10] if k != "probe"), end=" | ")
P
  grep "probe kernel" $R/gpurun_out/pl_$v.log | tail -1
  rm -rf $R/gpurun_out/pl_$v
done
