#!/usr/bin/env python3
"""Average per-dispatch PMC counter values per kernel from rocprofv3 --pmc CSV output (counter_collection.csv files under a directory)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if filt in k:
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-40s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
