cd "$GRAFT_REPO_ROOT"
export ST_BENCH_NO_KERNEL_TIMING=1
run() { lbl=$1; shift
  for b in $BATCHES; do
    env "$@" python bench.py --batch $b --steps 30 --warmup 4 --no-cpu-baseline --no-extras $SIZE 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('%-28s batch %3d: %7.0f frames/s  %.3f ms/step' % ('$lbl', $b, d['value'], d['ms_per_step']), flush=True)"
  done
}
SIZE="--height 2160 --width 3840" BATCHES="1 2 4 8" run 4k_auto A=1
SIZE="--height 2160 --width 3840" BATCHES="1 2 4 8" run 4k_noroles ST_ITER_ROLES=0
SIZE="--height 720 --width 1280" BATCHES="1 4 16" run 720p_auto A=1
SIZE="--height 720 --width 1280" BATCHES="1 4 16" run 720p_noroles ST_ITER_ROLES=0
SIZE="--height 480 --width 640" BATCHES="1 8 32" run 480p_auto A=1
SIZE="--height 480 --width 640" BATCHES="1 8 32" run 480p_noroles ST_ITER_ROLES=0
