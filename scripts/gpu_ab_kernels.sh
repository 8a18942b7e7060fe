#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Per-kernel-class times of one headline step for experimental library builds (scripts/build_variant.sh):
#   bash scripts/gpu_ab_kernels.sh default nt1 ...     ("default" = the production library)
set -u
cd "$GRAFT_REPO_ROOT"
for name in "$@"; do
  lib=""
  [ "$name" != "default" ] && lib="$PWD/scannertools_amd/lib_exp_$name/libscannertools_hip.so"
  ST_HIP_LIB=$lib timeout 300 python scripts/time_kernels.py 6 2>/dev/null | tail -1 | sed "s|^[^ ]*|$name|"
done
