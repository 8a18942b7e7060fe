cd "$GRAFT_REPO_ROOT"
export ST_BENCH_NO_KERNEL_TIMING=1
bash scripts/trace_small.sh 256 4 > /dev/null; echo "== default"; grep "k_flow_iter\|step" gpurun_out/ts_256/timeline.txt
ST_ITER_ROLES=1 ST_ROLES_NCW=4 bash scripts/trace_small.sh 256 4 > /dev/null; echo "== roles4"; grep "k_flow_iter\|step" gpurun_out/ts_256/timeline.txt
ST_ITER_ROLES=1 ST_ROLES_NCW=5 bash scripts/trace_small.sh 256 4 > /dev/null; echo "== roles5"; grep "k_flow_iter\|step" gpurun_out/ts_256/timeline.txt
