for s in 11388 11461; do echo -n "default seed $s: "; timeout -k 10 120 python3 tests/diag/pipeline_stress.py $s $((s+1)) 2>&1 | tail -1; done
