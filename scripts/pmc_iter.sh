#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Counter passes restricted to the flow-iteration kernel (fast: other kernels are not instrumented).
# usage: bash scripts/pmc_iter.sh <tag> [lib variant name | default] [kernel regex]
set -u
tag=$1; var=${2:-default}; rx=${3:-k_flow_iter3}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
[ "$var" != "default" ] && export ST_HIP_LIB="$GRAFT_REPO_ROOT/scannertools_amd/lib_exp_$var/libscannertools_hip.so"
out=gpurun_out/pi_$tag
mkdir -p $out
pass() { name=$1; shift
  timeout 180 rocprofv3 --kernel-include-regex "$rx" --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass p1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
pass p2 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass p3 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass p4 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
pass p5 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass p6 TCP_TCC_READ_REQ_LATENCY_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum
python3 scripts/pmc_summary.py $out > $out/summary.txt 2>&1
for p in p1 p2 p3 p4 p5 p6; do tail -2 $out/$p.log > $out/$p.tail; rm -rf $out/$p $out/$p.log; done
