#!/bin/bash
for v in "$@"; do echo "== $v"; SAGE2OV_TIMING=1 SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 400 python3 tests/diag/noisy_phases.py 10000000 1000 2>&1 | grep -E "reduce/" | tail -2; done
