#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# small-batch sweep over an environment knob: bash scripts/gpu_small_sweep.sh VAR "v1 v2 ..." "batches"
cd "$GRAFT_REPO_ROOT"
var=$1; vals=$2; batches=${3:-"1 2 4 8"}
for v in $vals; do
  for b in $batches; do
    env $var=$v python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$var=$v batch %3d: %7.0f frames/s  %.3f ms/step' % ($b, d['value'], d['ms_per_step']), flush=True)"
  done
done
