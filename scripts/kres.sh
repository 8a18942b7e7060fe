#!/bin/bash
# Register / LDS use of the kernels of one csrc file: bash scripts/kres.sh st_farneback [filter] [extra flags]
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
f=$1; flt=${2:-}; shift; [ $# -gt 0 ] && shift
tmp=$(mktemp -d)
cd "$tmp"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden \
  -I"$root/include" -I"$root/scannertools_amd/csrc" -save-temps=obj "$@" -c "$root/scannertools_amd/csrc/$f.hip" -o "$tmp/f.o" 2>/dev/null
python3 "$root/scripts/kernel_resources.py" "$tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s" "$flt"
rm -rf "$tmp"
