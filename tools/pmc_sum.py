"""Sums rocprofv3 --pmc counter rows per kernel: prints counter totals for kernels matching a substring."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); calls = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    print("%-28s %.6g  (dispatches %d)" % (k, acc[k], len(calls[k])))
