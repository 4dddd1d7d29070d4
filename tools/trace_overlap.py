#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: wall span, union of busy time, sum of kernel durations
(sum / union = average number of kernels in flight), per-kernel totals."""
import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # drop the first N kernels (warm-up)
rows = rows[skip:]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print("kernels %d  span %.3f ms  busy(union) %.3f ms  sum of durations %.3f ms  avg in flight %.2f" % (len(rows), span / 1e6, busy / 1e6, tot / 1e6, tot / busy))
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", k); k = re.sub(r"^void ", "", k); k = re.split(r"[(<]", k)[0].split("::")[-1][:40]
    per[k][0] += 1; per[k][1] += e - s
for k, (n, t) in sorted(per.items(), key=lambda x: -x[1][1])[:12]:
    print("  %-40s n=%5d  total %.3f ms  mean %.1f us" % (k, n, t / 1e6, t / n / 1e3))
