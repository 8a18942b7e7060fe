#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a gfx950 assembly dump
(hipcc ... -save-temps=obj -> *-hip-amdgcn-amd-amdhsa-gfx950.s).  Usage: kernel_resources.py file.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in s.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if flt not in name:
        continue

    def g(k):
        m = re.search(r"\." + k + r":\s+(\d+)", b)
        return m.group(1) if m else "?"
    print("%-60s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s" % (
        name[:60], g("vgpr_count"), b.split("\n")[0].strip(), g("sgpr_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
