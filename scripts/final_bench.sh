#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# The round's committed bench line: bash scripts/final_bench.sh <tag>   (run on the GPU box; copy gpurun_out/bench_<tag>.json to profiles/<tag>_bench.json)
cd "$GRAFT_REPO_ROOT"
tag=${1:-final}
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -1 gpurun_out/bench_$tag.err
python - "$tag" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_%s.json" % sys.argv[1]))
print(d["value"], d["ms_per_step"], {k: d["roofline"][k] for k in ("frac", "frac_model_8d", "frac_traffic", "traffic", "l2_hit_rate")})
e = d["extra"]
for k, v in e["optical_flow_small_batches"].items():
    if isinstance(v, dict):
        print(k, {a: round(b, 1) for a, b in v.items()})
print({k: (round(v["frames_per_s"]) if isinstance(v, dict) and "frames_per_s" in v else "") for k, v in e.items()})
PY
