"""Summarise a tools/profile_gpu.sh output directory: per-kernel time from the
kernel trace and per-kernel sums of every PMC counter collected."""
import csv, glob, os, sys, collections

out = sys.argv[1]

import re

def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:60]

for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print(f"  {short(row['Name']):<44} calls {row['Calls']:>6} total_ns {row['TotalDurationNs']:>14} avg_ns {row['AverageNs']:>14} pct {row['Percentage']}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    if sums:
        print("== pmc:", os.path.basename(d))
        for k, cs in sums.items():
            print(f"  {k:<44} dispatches {len(calls[k]):>5} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(cs.items())))
