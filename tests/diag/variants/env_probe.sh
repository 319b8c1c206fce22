#!/bin/bash
# usage: [READS=50000000] env_probe.sh "ENV=1 ..." ...   -> probe pass alone (tests/diag/probe_only.py) of the product library under each set of environment switches
for e in "$@"; do echo -n "[$e] "; env $e timeout -k 10 300 python3 tests/diag/probe_only.py ${READS:-10000000} 3 2>&1 | tail -1; done
