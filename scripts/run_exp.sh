cd "$GRAFT_REPO_ROOT"
source scripts/exp_small2.sh
BATCHES="1 2 4 8" run single0 ST_POLY_SINGLE_MAX=0
BATCHES="1 2 4 8" run single8 ST_POLY_SINGLE_MAX=8
