#!/usr/bin/env python3
"""Timeline of ONE step out of a rocprofv3 rocpd database (kernel-trace): every kernel between two consecutive launches
of the step's first kernel, with its start relative to the step, its duration, the gap since the previous kernel's end
and its grid -- what a per-kernel stats table hides (gaps, overlap between streams, which launch is which level).
   python scripts/rocpd_timeline.py x_results.db [first_kernel_substring] [step_index]
(no step index: the step of median duration)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_hist"
which = int(sys.argv[3]) if len(sys.argv) > 3 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in ("name", "start", "end", "grid_x", "grid_y", "grid_z", "workgroup_x", "stream_id", "queue_id") if c in cols]
rows = db.execute("select %s from kernels order by start" % ", ".join(want)).fetchall()
idx = {c: i for i, c in enumerate(want)}
starts = [i for i, r in enumerate(rows) if first in r[idx["name"]]]
if len(starts) < 3:
    sys.exit("fewer than 3 launches of %r" % first)
if which is None:
    # the step of median duration among the steady ones (a step that met a host hiccup shows gaps that are not the kernels')
    cand = list(range(len(starts) // 4, len(starts) - 1))
    dur = sorted((rows[starts[i + 1]][idx["start"]] - rows[starts[i]][idx["start"]], i) for i in cand)
    which = dur[len(dur) // 2][1]
a, b = starts[which], starts[which + 1]
t0 = rows[a][idx["start"]]
prev_end = None
print("%-44s %9s %9s %8s %-18s %s" % ("kernel", "start_us", "dur_us", "gap_us", "grid(wgs)", "queue"))
busy = 0
for r in rows[a:b]:
    name = r[idx["name"]]
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    short = (m.group(1) if m else name.split("(")[0])[-44:]
    s, e = r[idx["start"]], r[idx["end"]]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    wg = r[idx["workgroup_x"]] if "workgroup_x" in idx else 1
    g = "%dx%dx%d" % (r[idx["grid_x"]] // max(wg, 1), r[idx["grid_y"]], r[idx["grid_z"]]) if "grid_x" in idx else ""
    q = r[idx["queue_id"]] if "queue_id" in idx else (r[idx["stream_id"]] if "stream_id" in idx else "")
    print("%-44s %9.1f %9.1f %8.1f %-18s %s" % (short, (s - t0) / 1e3, (e - s) / 1e3, gap, g, q))
    prev_end = max(prev_end, e) if prev_end is not None else e
    busy += e - s
print("step: %.1f us wall, %.1f us of kernel time (sum over streams)" % ((rows[b][idx["start"]] - t0) / 1e3, busy / 1e3))
