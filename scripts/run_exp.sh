cd "$GRAFT_REPO_ROOT"
source scripts/exp_small2.sh
export STEPS=30
BATCHES="2 4 16 32 64" run default A=1
BATCHES="2 4 16 32 64" run roles ST_ITER_ROLES=1
BATCHES="2 4 16 32" run roles5 ST_ITER_ROLES=1 ST_ROLES_NCW=5
BATCHES="2 4 16 32" run roles4 ST_ITER_ROLES=1 ST_ROLES_NCW=4
