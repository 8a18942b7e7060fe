#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# tile-kernel threshold sweep at mid batch sizes: bash scripts/gpu_tilepx_sweep.sh
cd "$GRAFT_REPO_ROOT"
for b in 4 8 16 32 64; do
  for px in 600000 1200000 2500000 5000000 10000000; do
    ST_ITER_TILE_PX=$px python bench.py --batch $b --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('batch %3d tile_px %9d: %7.0f frames/s  %.3f ms/step' % ($b, $px, d['value'], d['ms_per_step']), flush=True)"
  done
done
