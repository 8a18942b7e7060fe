#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Records the tolerance tier of every flow field of tests/test_fuzz_gpu.py::test_fuzz_optical_flow (run on a GPU box):
#   bash scripts/record_flow_tiers.sh > tests/golden/flow_fuzz_tiers.json
cd "$GRAFT_REPO_ROOT"
ST_RECORD_FLOW_TIERS=1 python -m pytest tests/test_fuzz_gpu.py -q -m gpu -k "test_fuzz_optical_flow and default" -s 2>/dev/null | python -c "
import json, sys, ast
out = {}
for ln in sys.stdin:
    if 'FLOW_TIERS_RECORD' in ln:
        p = ln[ln.index('FLOW_TIERS_RECORD'):].split(None, 2)
        out[p[1]] = ast.literal_eval(p[2].strip())
print(json.dumps(out, indent=0, sort_keys=True))"
