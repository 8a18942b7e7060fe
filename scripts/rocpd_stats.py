#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table:
   python scripts/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/xxx.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-64s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for name, n, s, a, mn, mx in rows:
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    short = m.group(1) if m else name.split("<")[0][-64:]
    print("%-64s %7d %12.1f %12.2f %12.2f %12.2f %6.2f" % (short, n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
