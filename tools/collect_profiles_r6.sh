#!/bin/bash
# runs on the GPU box from the repo root: the round-6 evidence in one call -> gpurun_out/prof_r06/
#   1. the headline mode ('fp32_storage'): kernel stats + timelines (bs 8, bs 1), SQ counters, HBM PMC (tools/collect_profiles_p2.sh)
#   2. the 'fp16' mode's kernels (unchanged since round 4): kernel stats (tools/collect_profiles.sh does the PMC sets; here stats only)
#   3. socket power + clock per kernel class and per whole step (tools/power_trace_r6.py)
#   4. the training iteration's kernel stats
#   5. the default bench line
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
tools/collect_profiles_p2.sh $TAG > $OUT/collect_p2.log 2>&1
python tools/pmc_to_bench_p2.py $OUT/${TAG}_precise_pmc_hbm_traffic_raw.json $OUT/pmc_traffic_precise.json > /dev/null 2>&1
cd /tmp; rm -rf /tmp/p_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -o kt -- python $R/bench.py --steps 30 --warmup 5 --pipeline 1 --headline-mode fp16 --no-cpu-baseline --no-latency --no-train --no-siblings --no-configs --sustained-s 0 > $OUT/${TAG}_fp16_bench_under_rocprof.json 2>/tmp/kt.err
find /tmp/p_kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_fp16_kernel_stats.csv \;
cd $R
python tools/power_trace_r6.py 2.0 > $OUT/${TAG}_power_trace.log 2>&1; cp gpurun_out/power_trace_r6.json $OUT/${TAG}_power_trace.json
tools/timing/train_prof.sh $TAG > $OUT/train_prof.log 2>&1; cp gpurun_out/${TAG}_train_step_kernel_stats.csv $OUT/ 2>/dev/null
python bench.py > $OUT/${TAG}_bench_default.json 2>$OUT/bench.err
ls -la $OUT
