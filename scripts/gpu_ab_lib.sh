#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# A/B of experimental library builds (scripts/build_variant.sh): bash scripts/gpu_ab_lib.sh default nt1 nt3 ...
# "default" = the production library; every other name = scannertools_amd/lib_exp_<name>.  Optional env BENCH_ARGS.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/abl
# The .so files do not travel to the GPU box (.gpurunignore): the production library is built here first, and a variant given
# as name=FLAGS (e.g. abl128=-DST_ABLATE=128) is built on the spot when its library is missing.
python -c "import __graft_entry__ as g; g.ensure_built()"
for item in "$@"; do
  name="${item%%=*}"
  lib=""
  if [ "$name" != "default" ]; then
    lib="$GRAFT_REPO_ROOT/scannertools_amd/lib_exp_$name/libscannertools_hip.so"
    if [ ! -f "$lib" ]; then
      flags=""; [ "$item" != "$name" ] && flags="${item#*=}"
      bash scripts/build_variant.sh "$name" "$flags" > /dev/null || { echo "$name BUILD FAILED"; continue; }
    fi
  fi
  ST_HIP_LIB=$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/abl/$name.json 2> gpurun_out/abl/$name.err
  python - "$name" gpurun_out/abl/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-24s fps %.0f  ms/step %.2f  iter avg ms %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]), flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
done
