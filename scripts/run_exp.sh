cd "$GRAFT_REPO_ROOT"
for n in 32 64 128; do for c in 0 1 2 4 8 16; do echo "N=$n chunks=$c: $(N=$n ST_HIST_CHUNKS=$c python scripts/bench_hist.py 2>/dev/null | grep 'bins 256' | tr '\n' '|')"; done; done
