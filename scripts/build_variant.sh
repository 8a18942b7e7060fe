#!/bin/bash
# Experimental build of the HIP library with extra -D flags on st_farneback.hip:
#   bash scripts/build_variant.sh <name> "-DST_EXP_NT=1 ..."   ->  scannertools_amd/lib_exp_<name>/libscannertools_hip.so
# (select it with ST_HIP_LIB=<that path>; the directories are git-ignored and travel with gpurun)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/scannertools_amd/lib_exp_$name
mkdir -p $out
make -C $root/scannertools_amd/csrc -j4 > /dev/null
cd $root/scannertools_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -I../../include -Wall -Wno-unused-function $flags -c st_farneback.hip -o $out/st_farneback.o
objs=""
for f in st_context st_hist st_flowvis st_imgproc st_pose st_conv st_conv_tile_bf16x3 st_conv_tile_f32; do objs="$objs ../lib/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libscannertools_hip.so $out/st_farneback.o $objs -Wl,-rpath,/opt/rocm/lib
rm -f $out/st_farneback.o
echo built $out/libscannertools_hip.so
