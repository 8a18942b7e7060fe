cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_flow_gpu.py -x -q -m gpu 2>&1 | tail -2
source scripts/exp_small2.sh
BATCHES="1 2 4 8 16" run auto A=1
