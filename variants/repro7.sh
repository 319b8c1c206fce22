export SAGE2OV_DEBUG_HITS=1
for rep in 1 2 3; do for v in head tree; do echo -n "$v: "; if [ $v = tree ]; then unset SAGE2OV_LIB; else export SAGE2OV_LIB=$PWD/variants/libsage2ov_$v.so; fi; timeout -k 10 120 python3 tests/diag/pipeline_stress.py 11461 11462 2>&1 | grep -E "hits|counters" | cut -c1-230 | tr '\n' ' '; echo; done; done
