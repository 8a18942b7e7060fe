#!/usr/bin/env python3
"""Average the per-dispatch counters of a scripts/pmc_profile.sh run per kernel:
   python scripts/pmc_summary.py gpurun_out/pmc_<tag> [> profiles/...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
            if not m or "at::" in name:
                continue
            k = m.group(1)
            # distinguish launches of the same kernel by grid size (pyramid level)
            key = "%s[g=%s]" % (k, row.get("Grid_Size", "?"))
            a = acc[key][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for key in sorted(acc):
    print(key)
    for c in sorted(acc[key]):
        s, n = acc[key][c]
        print("    %-28s avg/dispatch %16.1f   (n=%d)" % (c, s / n, n))
