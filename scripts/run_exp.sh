cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_flow_gpu.py -x -q -m gpu 2>&1 | tail -2
source scripts/exp_small2.sh
BATCHES="1 2 8" run auto A=1
export ST_BENCH_NO_KERNEL_TIMING=1; bash scripts/trace_small.sh 1 40 > /dev/null; grep "tile\|step" gpurun_out/ts_1/timeline.txt
