#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Kernel trace of small-batch steps: bash scripts/trace_small.sh <batch> [steps]
# prints the per-kernel stats table and the timeline of one step (scripts/rocpd_timeline.py)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
b=$1; steps=${2:-40}
out=gpurun_out/ts_$b; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --batch $b --steps $steps --warmup 3 --no-cpu-baseline --no-extras > $out/bench.json 2> $out/err.log
python3 scripts/rocpd_stats.py $out/trace/trace_results.db | grep -v "at::native\|rocclr\|distribution" > $out/stats.txt
python3 scripts/rocpd_timeline.py $out/trace/trace_results.db k_gray > $out/timeline.txt 2>&1
rm -rf $out/trace
cat $out/stats.txt $out/timeline.txt
python3 -c "
import json;d=json.load(open('$out/bench.json'));print('batch $b: %.0f frames/s %.3f ms/step'%(d['value'],d['ms_per_step']))"
