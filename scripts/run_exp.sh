cd "$GRAFT_REPO_ROOT"
source scripts/exp_small2.sh
for r in 8 12 16 24 36 48; do BATCHES="1 2 8" STEPS=60 run pe_minrows_$r ST_PE_MINROWS=$r; done
