cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_flow_gpu.py -x -q -m gpu 2>&1 | tail -15
source scripts/exp_small2.sh
BATCHES="1 8" run default A=1
BATCHES="1 8" run roles ST_ITER_ROLES=1
BATCHES="1 8" run roles4 ST_ITER_ROLES=1 ST_ROLES_NCW=4
BATCHES="1 8" run roles5 ST_ITER_ROLES=1 ST_ROLES_NCW=5
BATCHES="256" STEPS=6 run big_default A=1
BATCHES="256" STEPS=6 run big_roles ST_ITER_ROLES=1 
BATCHES="256" STEPS=6 run big_roles4 ST_ITER_ROLES=1 ST_ROLES_NCW=4
