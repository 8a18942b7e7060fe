cd "$GRAFT_REPO_ROOT"
export ST_BENCH_NO_KERNEL_TIMING=1 ST_PYR_ROLES=1
for r in 216 184 160 136 120 96 64; do
  ST_PYR_SEGROWS=$r bash scripts/trace_small.sh 256 3 > /dev/null; echo "rows $r: $(grep k_pyr gpurun_out/ts_256/timeline.txt)"
done
