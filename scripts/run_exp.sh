cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_flow_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -2
export ST_BENCH_NO_KERNEL_TIMING=1
bash scripts/trace_small.sh 256 4 > /dev/null; grep "k_flow_iter\|step" gpurun_out/ts_256/timeline.txt
unset ST_BENCH_NO_KERNEL_TIMING
python bench.py --no-cpu-baseline --no-extras --steps 10 | python -c "
import json,sys
d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --no-extras --steps 10 | python -c "
import json,sys
d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
source scripts/exp_small2.sh
BATCHES="1 8" run auto A=1
