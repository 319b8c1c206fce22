#!/bin/bash
for v in "$@"; do echo -n "$v: "; SAGE2OV_LIB=$PWD/tests/diag/variants/build/libsage2ov_$v.so timeout -k 10 200 python3 tests/diag/probe_only.py 10000000 3 2>&1 | tail -1; done
