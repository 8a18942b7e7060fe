#!/usr/bin/env python3
"""Per-kernel HBM traffic from a scripts/profile_round.sh run.
   python scripts/pmc_traffic.py gpurun_out/round_<tag> [steps]
Sums each counter over all dispatches of a kernel and divides by the dispatch count.
READ bytes: TCC_EA0_RDREQ counts requests of 32/64/128 B; FETCH_SIZE (KB) tallies every request
at 64 B and therefore under-reports 128-B requests by 2x on gfx950 (MI355X_MICROARCH.md, HBM).
WRITE_SIZE (KB) matched known byte counts exactly on this pool (k_update_matrices: 20 B/px)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(k_[a-z0-9_]+)", row.get("Kernel_Name", ""))
            if not m or "at::" in row["Kernel_Name"]:
                continue
            acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[m.group(1)][row["Counter_Name"]] += 1
res = {}
for k in sorted(acc):
    a, c = acc[k], cnt[k]
    avg = {n: a[n] / c[n] for n in a}
    n32, n64, n128 = avg.get("TCC_EA0_RDREQ_32B_sum", 0), avg.get("TCC_EA0_RDREQ_64B_sum", 0), avg.get("TCC_EA0_RDREQ_128B_sum", 0)
    tot = avg.get("TCC_EA0_RDREQ_sum", 0)
    rd_sized = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(tot - n32 - n64 - n128, 0)
    entry = {
        "launches_sampled": max(c.values()),
        "FETCH_SIZE_bytes": avg.get("FETCH_SIZE", 0) * 1024,
        "WRITE_SIZE_bytes": avg.get("WRITE_SIZE", 0) * 1024,
        "RDREQ": tot, "RDREQ_32B": n32, "RDREQ_64B": n64, "RDREQ_128B": n128,
        "read_bytes_by_request_size": rd_sized,
        "hbm_bytes_per_launch": rd_sized + avg.get("WRITE_SIZE", 0) * 1024,
        "L2_hit_rate": avg.get("TCC_HIT_sum", 0) / max(avg.get("TCC_HIT_sum", 0) + avg.get("TCC_MISS_sum", 0), 1),
    }
    for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
        if n in avg:
            entry[n] = avg[n]
    res[k] = entry
# which kernel sources the counters belong to: bench.py reports `traffic` only while they still match
import hashlib
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scannertools_amd", "csrc")
# (st_internal.h and the Makefile shape the kernels' code generation too: st_gl's address spaces, -fno-slp-vectorize)
res["_meta"] = {"source_sha256": {f: hashlib.sha256(open(os.path.join(src, f), "rb").read()).hexdigest()
                                  for f in ("st_farneback.hip", "st_hist.hip", "st_internal.h", "Makefile")}}
print(json.dumps(res, indent=1))
