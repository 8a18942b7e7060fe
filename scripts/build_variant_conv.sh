#!/bin/bash
# like build_variant.sh, for st_conv.hip:  bash scripts/build_variant_conv.sh <name> "<flags>"
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/scannertools_amd/lib_exp_$name
mkdir -p $out
make -C $root/scannertools_amd/csrc -j4 > /dev/null
cd $root/scannertools_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -I../../include -Wall -Wno-unused-function $flags -c st_conv.hip -o $out/st_conv.o
objs=""
for f in st_context st_hist st_flowvis st_imgproc st_pose st_farneback st_conv_tile_bf16x3 st_conv_tile_f32; do objs="$objs ../lib/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libscannertools_hip.so $out/st_conv.o $objs -Wl,-rpath,/opt/rocm/lib
rm -f $out/st_conv.o
echo built $out/libscannertools_hip.so
