export SAGE2OV_PROBE_TAIL=0
for v in cut2 cut5; do echo -n "$v reuse: "; SAGE2OV_LIB=$PWD/variants/libsage2ov_$v.so timeout -k 10 200 python3 tests/diag/probe_only.py 10000000 3 2>&1 | tail -1; echo -n "$v no-reuse: "; SAGE2OV_NO_WINDOW_REUSE=1 SAGE2OV_LIB=$PWD/variants/libsage2ov_$v.so timeout -k 10 200 python3 tests/diag/probe_only.py 10000000 3 2>&1 | tail -1; done
echo -n "full no-reuse: "; SAGE2OV_NO_WINDOW_REUSE=1 timeout -k 10 200 python3 tests/diag/probe_only.py 10000000 3 2>&1 | tail -1
