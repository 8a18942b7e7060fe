#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Latency / front-end view of the flow-iteration kernel (separate counter-only passes; pmc_summary.py averages per
# kernel and grid size).  usage: bash scripts/pmc_lat.sh <tag> [extra bench args]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/lat_$tag
mkdir -p $out
BENCH_ARGS=("$@")
rocprofv3 -L > $out/avail.txt 2>&1
pass() { name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "${BENCH_ARGS[@]}" > $out/$name.log 2>&1
}
pass a SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
pass b TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass c TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pass d TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass e SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass f TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
python3 scripts/pmc_summary.py $out > $out/summary.txt 2>&1
for p in a b c d e f; do tail -3 $out/$p.log > $out/$p.tail; rm -rf $out/$p $out/$p.log; done
grep -c . $out/summary.txt
