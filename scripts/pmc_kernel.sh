#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Wider counter set for one kernel of the bench: bash scripts/pmc_kernel.sh <tag> <kernel regex>
set -u
tag=$1; rx=$2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pk_$tag
mkdir -p $out
pass() { name=$1; shift
  timeout 180 rocprofv3 --kernel-include-regex "$rx" --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass p1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass p2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
pass p3 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
pass p4 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass p5 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_WRITE_sum
pass p6 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA
python3 scripts/pmc_summary.py $out > $out/summary.txt 2>&1
for p in p1 p2 p3 p4 p5 p6; do rm -rf $out/$p $out/$p.log; done
cat $out/summary.txt
