#!/usr/bin/env python3
"""Sum every counter of a rocprofv3 --pmc run per kernel: python scripts/pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if len(sys.argv) > 2 and sys.argv[2] not in kn:
            continue
        kn = kn.split("(")[0][-60:]
        acc[kn][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[kn][row["Counter_Name"]] += 1
for kn in sorted(acc):
    print(kn)
    for c in sorted(acc[kn]):
        print("   %-32s %16.0f per launch (%d launches)" % (c, acc[kn][c] / cnt[kn][c], cnt[kn][c]))
