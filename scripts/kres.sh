#!/bin/bash
# Register / LDS use of the kernels of one csrc file: bash scripts/kres.sh st_farneback [filter] [extra flags]
f=$1; flt=$2; shift; shift
mkdir -p /tmp/st_kres && cd /tmp/st_kres && rm -f *.s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden \
  -I/root/repo/include -I/root/repo/scannertools_amd/csrc -save-temps=obj "$@" -c /root/repo/scannertools_amd/csrc/$f.hip -o /tmp/st_kres/f.o 2>/dev/null
python3 /root/repo/scripts/kernel_resources.py /tmp/st_kres/$f-hip-amdgcn-amd-amdhsa-gfx950.s "$flt"
