cd "$GRAFT_REPO_ROOT"
t() { s=$(date +%s.%N); "$@" > /dev/null 2>&1; e=$(date +%s.%N); echo "$(echo "$e - $s" | bc) s: $*"; }
t python bench.py --no-extras --no-cpu-baseline
t python bench.py --no-extras
t python bench.py --no-cpu-baseline --no-4k --no-pose --no-shots
t python bench.py --no-cpu-baseline --no-4k --no-shots
t python bench.py
