cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_imgproc_gpu.py tests/test_flowvis_gpu.py tests/test_pose_net_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-4k --no-pose --no-shots --steps 4 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -2 gpurun_out/bench_now.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_now.json"))
e=d["extra"]
for k,v in e["optical_flow_small_batches"].items():
    if isinstance(v,dict): print(k, {a:round(b,1) for a,b in v.items()})
print(json.dumps(e["host_fed"],indent=0)[:900])
PY
