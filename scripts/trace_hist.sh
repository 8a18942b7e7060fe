#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Kernel-trace durations of the Histogram kernel at N frames per launch beside its HIP-event figure:
#   bash scripts/trace_hist.sh <tag> [N ...]      (on the GPU box; writes gpurun_out/th_<tag>_<N>.txt)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for n in "${@:-32 64}"; do
  out=gpurun_out/th_${tag}_$n
  rm -rf $out; mkdir -p $out
  N=$n timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 scripts/bench_hist.py > $out.log 2>&1
  python3 - $out $n <<'PY' > gpurun_out/th_${tag}_$n.txt
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
by = {}
for r in rows:
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
    by.setdefault(m.group(1) if m else r["Kernel_Name"].split("(")[0][-60:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-62s n=%4d  median %8.2f us  min %8.2f  max %8.2f" % (k, len(v), v[len(v) // 2], v[0], v[-1]))
PY
  cat gpurun_out/th_${tag}_$n.txt | head -8; grep "bins" $out.log
  rm -rf $out
done
