export SAGE2OV_PROBE_TAIL=0
echo -n "default: "; timeout -k 10 200 python3 tests/diag/probe_only.py 10000000 3 2>&1 | tail -1
echo -n "no minimiser index: "; SAGE2OV_NO_MINIMIZER_INDEX=1 timeout -k 10 200 python3 tests/diag/probe_only.py 10000000 3 2>&1 | tail -1
