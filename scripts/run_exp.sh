cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -2
SCANNERTOOLS_FLOW_COPIES=overlap timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "OpticalFlow" 2>&1 | tail -2
for m in serial overlap serial overlap; do
SCANNERTOOLS_FLOW_COPIES=$m python bench.py --no-cpu-baseline --no-4k --no-pose --no-shots --steps 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); e=d['extra']; print('$m', {k:round(v,1) for k,v in e['host_fed']['OpticalFlow'].items()}, round(e['legacy_flow_hist']['host_fed']['OpticalFlow_426x240_frames_per_s']))"
done
