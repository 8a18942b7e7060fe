#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Convolution kernel: per-layer timing, then counter passes over one layer shape (default: the stages' 7x7 128->128).
#   bash scripts/gpu_conv.sh [H W CIN COUT K]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/conv; mkdir -p $out
shape="${*:-46 82 128 128 7}"
N=16 timeout 300 python3 scripts/bench_conv_layers.py 2>&1 | tail -17
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  rm -rf $out/pmc; N=32 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/pmc -o p -- python3 scripts/conv_one_layer.py $shape > $out/pmc.log 2>&1 || tail -3 $out/pmc.log
  echo "-- $grp"; python3 scripts/pmc_sum.py $out/pmc k_conv; grep TFLOP $out/pmc.log
done
rm -rf $out/pmc
