for mode in "SAGE2OV_PROBE_SAMPLE_MIN=1024" "SAGE2OV_PROBE_SAMPLE_MIN=1024 SAGE2OV_NO_PREHITS=1" "SAGE2OV_PROBE_TAIL=2" "SAGE2OV_PROBE_TAIL=2 SAGE2OV_NO_PREHITS=1" "SAGE2OV_PROBE_SAMPLE_MIN=1024 SAGE2OV_HOST_REDUCE=1"; do
  for s in 11388 11461; do echo -n "$mode seed $s: "; env $mode timeout -k 10 120 python3 tests/diag/pipeline_stress.py $s $((s+1)) 2>&1 | tail -1; done
done
