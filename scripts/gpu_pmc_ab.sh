#!/bin/bash
# A/B of k_flow_iter schedules under the PMC counters: ST_PAIRS_PER_WG=1 vs 2 (traffic, L2 hit rate).
#   bash scripts/gpu_pmc_ab.sh <tag>      (on the GPU box, from the repo root)
set -u
tag=${1:-ab}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for np in 1 2; do
  export ST_PAIRS_PER_WG=$np
  out=gpurun_out/pmc_${tag}_np$np
  mkdir -p $out
  pass() { name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.log 2>&1
  }
  pass write WRITE_SIZE
  pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  pass hit TCC_HIT_sum TCC_MISS_sum
  python3 scripts/pmc_traffic.py $out > $out/pmc_traffic.json
  rm -rf $out/write $out/rdreq $out/hit
  python3 - <<PY
import json
d=json.load(open("$out/pmc_traffic.json"))["k_flow_iter"]
print("NP=$np  bytes/launch %.3f GB  read %.3f GB  write %.3f GB  L2 hit %.3f" % (d["hbm_bytes_per_launch"]/1e9, d["read_bytes_by_request_size"]/1e9, d["WRITE_SIZE_bytes"]/1e9, d["L2_hit_rate"]))
PY
done
