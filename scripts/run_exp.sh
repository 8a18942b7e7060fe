cd "$GRAFT_REPO_ROOT"
source scripts/exp_small2.sh
BATCHES="2 4 8 16" run rb2 ST_ROLES_NMS=1
BATCHES="2 4 8 16" run rb4 ST_ROLES_NMS=14
export ST_BENCH_NO_KERNEL_TIMING=1 ST_ROLES_NMS=14; bash scripts/trace_small.sh 8 20 > /dev/null; grep "roles" gpurun_out/ts_8/timeline.txt
ST_ROLES_NMS=14 timeout 600 python -m pytest tests/test_flow_gpu.py -x -q -m gpu -k "schedules or iteration_parity" 2>&1 | tail -3
