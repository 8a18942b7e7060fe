#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# bf16x3 convolution kernel: timing of a few layer shapes + counter passes on one (default 46 82 128 128 7)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/conv3; mkdir -p $out
export MATH=${MATH:-bf16x3}
for sh in "46 82 128 128 7" "92 164 256 256 3" "184 328 128 128 3" "368 656 64 64 3" "46 82 512 512 3" "46 82 128 512 1"; do N=32 REPS=5 python3 scripts/conv_one_layer.py $sh 2>&1 | tail -1; done
shape="${*:-46 82 128 128 7}"
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  rm -rf $out/pmc; N=32 timeout 300 rocprofv3 --kernel-include-regex "k_conv" --pmc $grp --output-format csv -d $out/pmc -o p -- python3 scripts/conv_one_layer.py $shape > $out/pmc.log 2>&1 || tail -3 $out/pmc.log
  echo "-- $grp"; python3 scripts/pmc_sum.py $out/pmc k_conv
done
rm -rf $out/pmc
