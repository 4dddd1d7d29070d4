#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name (substring filter) and print per-dispatch means."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nn_cull")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        short = k.split("(")[0][-40:]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"]); nd[short].add((f, r["Dispatch_Id"]))
for k, c in acc.items():
    n = len(nd[k]);  print(k, "dispatches(sum over passes):", n)
    for name, v in sorted(c.items()): print("   %-24s %.4g per dispatch" % (name, v / max(1, sum(1 for x in nd[k] ))))
