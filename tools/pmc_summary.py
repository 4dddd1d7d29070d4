#!/usr/bin/env python3
"""Per-dispatch means of rocprofv3 --pmc counter_collection CSVs, per counter, for kernels whose name contains a
substring; forward (even) and reverse (odd) launches of the pair are listed separately.
    python tools/pmc_summary.py gpurun_out/pmc_cull nn_cull"""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nn_cull")
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    for label, sel in (("all", ids), ("forward (1st, 3rd, ...)", ids[0::2]), ("reverse (2nd, 4th, ...)", ids[1::2])):
        if not sel: continue
        acc = collections.defaultdict(float)
        for i in sel:
            for k, v in rows[i].items(): acc[k] += v
        print("%s | %s: %d dispatches" % (f.split("/")[-3], label, len(sel)))
        for k in sorted(acc): print("    %-24s %.5g" % (k, acc[k] / len(sel)))
