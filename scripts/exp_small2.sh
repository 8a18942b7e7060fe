cd "$GRAFT_REPO_ROOT"
export ST_BENCH_NO_KERNEL_TIMING=1
run() { # label, env...
  lbl=$1; shift
  for b in $BATCHES; do
    env "$@" python bench.py --batch $b --steps ${STEPS:-80} --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('%-40s batch %3d: %7.0f frames/s  %.3f ms/step' % ('$lbl', $b, d['value'], d['ms_per_step']), flush=True)"
  done
}
