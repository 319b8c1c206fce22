#!/bin/bash
# where the probe kernel's wave cycles go: waiting vs issuing, by instruction class (three PMC passes over the probe pass alone, tests/diag/probe_only.py).  Run from the repo root on a GPU box.
set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_wait; mkdir -p $O
N=${READS:-50000000}
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o w -- python3 tests/diag/probe_only.py $N 1 > $O/a.log 2>&1
echo a done
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $O/b -o w -- python3 tests/diag/probe_only.py $N 1 > $O/b.log 2>&1
echo b done
timeout -k 10 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/c -o w -- python3 tests/diag/probe_only.py $N 1 > $O/c.log 2>&1
echo c done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_wait/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_probe_fast" in r["Kernel_Name"] and "true, 2, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()): print(f"{k:28s} {v:.4g}")
P
