#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# A/B of environment-selected kernel variants: bash scripts/gpu_ab_env.sh "VAR1=a VAR2=b" "VAR1=c" ...
# Each argument is one configuration (space-separated env assignments); prints the bench's key numbers.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/ab/$i.json 2> gpurun_out/ab/$i.err
  python - "$cfg" gpurun_out/ab/$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-40s fps %.0f  ms/step %.2f  iter avg ms %.4f" % (sys.argv[1] or "(default)", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
