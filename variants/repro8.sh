export SAGE2OV_DEBUG_HITS=1
SAGE2OV_DEBUG_HITS_FILE=$PWD/gpurun_out/hits_head.bin SAGE2OV_LIB=$PWD/variants/libsage2ov_head.so timeout -k 10 120 python3 tests/diag/pipeline_stress.py 11461 11462 > /dev/null 2>&1
SAGE2OV_DEBUG_HITS_FILE=$PWD/gpurun_out/hits_tree.bin timeout -k 10 120 python3 tests/diag/pipeline_stress.py 11461 11462 > /dev/null 2>&1
python3 - <<'PY'
import numpy as np
dt = np.dtype([("from","<u4"),("to","<u4"),("len","<i4"),("seq_hi","<u2"),("type","u1"),("pad","u1"),("seq","<u4")])
a = np.fromfile("gpurun_out/hits_head.bin", dtype=dt); b = np.fromfile("gpurun_out/hits_tree.bin", dtype=dt)
a = a[a["from"] != 0]; b = b[b["from"] != 0]
a = np.sort(a, order=["from","seq"]); b = np.sort(b, order=["from","seq"])
print(len(a), len(b))
n = min(len(a), len(b)); d = np.nonzero((a["to"][:n] != b["to"][:n]) | (a["len"][:n] != b["len"][:n]) | (a["type"][:n] != b["type"][:n]) | (a["from"][:n] != b["from"][:n]))[0]
print("differing", len(d))
for i in d[:12]: print(a[i], b[i])
PY
