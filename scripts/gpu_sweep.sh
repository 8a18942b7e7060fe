#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Batch-size sweep of the bench (frames/s through OpticalFlow + Histogram): bash scripts/gpu_sweep.sh
cd "$GRAFT_REPO_ROOT"
for b in 1 2 4 8 16 32 64 128 256; do
  python bench.py --batch $b --steps $(( b < 16 ? 40 : 10 )) --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('1080p batch %4d: %8.0f frames/s  %.3f ms/step' % ($b, d['value'], d['ms_per_step']))"
done
for b in 8 32 64; do
  python bench.py --batch $b --height 2160 --width 3840 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('4K    batch %4d: %8.0f frames/s  %.3f ms/step' % ($b, d['value'], d['ms_per_step']))"
done
