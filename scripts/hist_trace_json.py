#!/usr/bin/env python3
"""gpurun_out/th_<tag>_<N>.txt (scripts/trace_hist.sh) -> profiles/hist_small_trace.json, the committed kernel-trace durations
of the Histogram kernel at small launches that bench.py's `histogram_small_batches` record quotes beside its HIP-event
figures (guarded by the hash of st_hist.hip):   python scripts/hist_trace_json.py r4"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rec = {}
for n in (32, 64, 256):
    path = os.path.join(ROOT, "gpurun_out", "th_%s_%d.txt" % (tag, n))
    if not os.path.exists(path):
        continue
    for line in open(path):
        m = re.match(r"(k_hist_u8c3_v2<32, (1024, 0|256, 4), (?:true|false)>)\s+n=\s*(\d+)\s+median\s+([\d.]+) us\s+min\s+([\d.]+)", line)
        if m:
            rec.setdefault("batch_%d" % n, {})["bins_256" if "1024" in m.group(2) else "bins_16"] = {
                "kernel": m.group(1), "launches": int(m.group(3)), "median_us": float(m.group(4)), "min_us": float(m.group(5))}
rec["source"] = "profiles/%s_hist_small_trace.txt (rocprofv3 --kernel-trace, scripts/trace_hist.sh)" % tag
rec["sha256_st_hist_hip"] = hashlib.sha256(open(os.path.join(ROOT, "scannertools_amd", "csrc", "st_hist.hip"), "rb").read()).hexdigest()
json.dump(rec, open(os.path.join(ROOT, "profiles", "hist_small_trace.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
