#!/bin/bash
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"   # the repo root when not run through gpurun
# Round profile: the bench line, rocprofv3 --kernel-trace --stats of the SAME command, and PMC
# passes (separate runs, counters only) for HBM traffic.  Run on the GPU box from the repo root:
#   bash scripts/profile_round.sh <tag>
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round_$tag
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
# the headline region only (the CPU baseline and the extra records run other sizes / no kernels)
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --no-cpu-baseline --no-extras > $out/trace.log 2>&1
pass() { name=$1; shift
  # only this library's kernels are instrumented (the input generator's thousands of torch launches would otherwise
  # dominate the run time of a counter pass)
  rocprofv3 --kernel-include-regex "k_(flow_iter|polyexp|pyr|gray|hist)" --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/$name.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass ta TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
# summarise on the box (the raw rocprofv3 output exceeds what gpurun copies back) and drop the raw data
mkdir -p $out/summary
python3 scripts/rocpd_stats.py $out/trace/trace_results.db > $out/summary/kernel_stats.txt
python3 scripts/pmc_traffic.py $out > $out/summary/pmc_traffic.json
cp $out/bench.json $out/summary/bench.json
python3 scripts/pmc_summary.py $out > $out/summary/pmc_by_launch.txt
rm -rf $out/trace $out/fetch $out/write $out/rdreq $out/wrreq $out/sq $out/tcp $out/ta $out/valu
# small-batch calls (what a drop-in graph without batch= gets): kernel timeline of one step at 1 and 8 pairs per call,
# no event brackets
for b in 1 8; do
  ST_BENCH_NO_KERNEL_TIMING=1 bash scripts/trace_small.sh $b 30 > /dev/null 2>&1
  cp gpurun_out/ts_$b/timeline.txt $out/summary/small_${b}_timeline.txt
  cp gpurun_out/ts_$b/stats.txt $out/summary/small_${b}_kernel_stats.txt
done
# the legacy flow-histogram pipeline at 426x240 (old/histograms.py:63-78), 256 pairs per call
rocprofv3 --kernel-trace --stats -d $out/ltrace -o trace -- python3 scripts/bench_legacy.py --batches 256 --steps 4 > $out/summary/legacy.json 2> $out/legacy.err
python3 scripts/rocpd_stats.py $out/ltrace/trace_results.db | grep -v "at::native\|rocclr\|distribution\|elementwise\|vectorized" > $out/summary/legacy_kernel_stats.txt
python3 scripts/rocpd_timeline.py $out/ltrace/trace_results.db k_resize_linear > $out/summary/legacy_timeline.txt
rm -rf $out/ltrace
# the Histogram kernel at Scanner-sized launches: kernel-trace durations beside the HIP-event figures
# (afterwards, in the build container: python scripts/hist_trace_json.py $tag -> profiles/hist_small_trace.json)
bash scripts/trace_hist.sh $tag 32 64 256 > /dev/null 2>&1
{ for n in 32 64 256; do echo "== $n frames of 1080p per launch (scripts/trace_hist.sh: rocprofv3 --kernel-trace of scripts/bench_hist.py; 3 kinds of frames x 21 launches per instance) =="; grep k_hist gpurun_out/th_${tag}_$n.txt; echo "-- the library's own timing of the same launches WHILE TRACED (events on the dispatch; the tracer's interception adds ~5 us -- untraced they read the trace's figure, bench.py histogram_small_batches):"; grep bins gpurun_out/th_${tag}_$n.log; done; } > $out/summary/hist_small_trace.txt
cat $out/bench.json
