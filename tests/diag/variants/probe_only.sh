#!/bin/bash
# usage: [READS=50000000] probe_only.sh name...   -> probe pass alone per variant (tests/diag/probe_only.py)
for v in "$@"; do echo -n "$v: "; SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 300 python3 tests/diag/probe_only.py ${READS:-10000000} 3 2>&1 | tail -1; done
