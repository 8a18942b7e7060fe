#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Counters of the small-call kernels: bash scripts/pmc_small.sh <pairs per call> [kernel regex]
# (bench.py --batch <pairs>, event-free; separate --pmc passes, counters only)
set -u
b=$1; rx=${2:-"k_(flow_iter_roles|flow_iter_tile|pyr_roles|polyexp_ml)"}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ps_$b
mkdir -p $out
export ST_BENCH_NO_KERNEL_TIMING=1
pass() { name=$1; shift
  timeout 180 rocprofv3 --kernel-include-regex "$rx" --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass p1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass p2 SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA
python3 scripts/pmc_summary.py $out > $out/summary.txt 2>&1
for p in p1 p2; do rm -rf $out/$p $out/$p.log; done
cat $out/summary.txt
