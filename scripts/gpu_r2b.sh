set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 900 python bench.py > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; echo "bench rc $?"
tail -3 gpurun_out/r2b/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2b/bench.json"))
print("fps", d["value"], "ms", d["ms_per_step"])
print("parity", d.get("parity"))
print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","cores","host_cores","sample")})
print("extra", json.dumps(d.get("extra"), indent=1))
PY
bash scripts/gpu_pmc_ab.sh r2b
