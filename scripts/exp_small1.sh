cd "$GRAFT_REPO_ROOT"
run() { # label, env...
  lbl=$1; shift
  for b in 1 8; do
    env "$@" python bench.py --batch $b --steps 80 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('%-40s batch %3d: %7.0f frames/s  %.3f ms/step' % ('$lbl', $b, d['value'], d['ms_per_step']), flush=True)"
  done
}
run default A=1
run notiming ST_BENCH_NO_KERNEL_TIMING=1
run notiming+tile ST_BENCH_NO_KERNEL_TIMING=1 ST_ITER_TILE=1
run notiming+nooverlap ST_BENCH_NO_KERNEL_TIMING=1 ST_NO_OVERLAP=1
run notiming+abl4 ST_BENCH_NO_KERNEL_TIMING=1 ST_HIP_LIB=$GRAFT_REPO_ROOT/scannertools_amd/lib_exp_abl4/libscannertools_hip.so
