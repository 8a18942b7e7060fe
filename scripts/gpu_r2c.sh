cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_hist_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-4k --steps 3 > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; echo rc $?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c/bench.json"))
print(json.dumps(d["extra"]["host_fed"], indent=1))
PY
for sb in 4 16 32; do SCANNERTOOLS_HIST_SUBBATCH=$sb python bench.py --no-cpu-baseline --no-4k --steps 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['extra']['host_fed']; print('sub $sb', d['Histogram'], d['Histogram_pageable_source'])"; done
