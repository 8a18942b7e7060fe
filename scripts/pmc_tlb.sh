#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_tlb; mkdir -p $out
pass() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1
}
pass tlb1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum
pass tlb2 GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
pass tlb3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass tlb4 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
