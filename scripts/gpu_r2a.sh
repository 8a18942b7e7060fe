set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests/test_flow_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/r2a/pytest_flow.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2a/pytest_flow.log
tail -15 gpurun_out/r2a/pytest_flow.log
for np in 1 2; do
  ST_PAIRS_PER_WG=$np timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_np$np.json 2> gpurun_out/r2a/bench_np$np.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2a/bench_np$np.json"))
print("NP=$np fps", d["value"], "ms/step", d["ms_per_step"], "iter avg ms", d["roofline"]["avg_launch_ms"], "launches", d["roofline"]["launches"])
PY
done
