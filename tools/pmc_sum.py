"""Sum rocprofv3 --pmc counter_collection.csv files per kernel and counter (one launch = one row per counter)."""
import csv, glob, sys, collections
N = float(sys.argv[2]) if len(sys.argv) > 2 else 8502556.0
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"]); acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for (k, c), v in sorted(acc.items()):
    if "probe_fast" in k or (len(sys.argv) > 3 and sys.argv[3] in k):
        print(f"{k:40s} {c:28s} {v:14.5g} launches {cnt[(k, c)]:3d} per_launch_per_read {v / cnt[(k, c)] / N:10.3f}")
