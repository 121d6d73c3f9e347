#!/bin/bash
# runs on the GPU box from the repo root: per-kernel stats + the timeline of the last step of the 'fp32_storage' mode
# (tools/bench_precise.py, eager launches).  usage: tools/prof_p2.sh <tag>
TAG=${1:-p2}
BS=${2:-8}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; rm -rf /tmp/p_kp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kp -o kp -- python $R/tools/bench_precise.py 10 $BS > $OUT/${TAG}_precise_bench_under_rocprof.json 2>/tmp/kp.err
find /tmp/p_kp -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_precise_kernel_stats.csv \;
T=$(find /tmp/p_kp -name "*kernel_trace.csv" | head -1)
python $R/tools/timing/timeline.py $T k_pl_stem > $OUT/${TAG}_precise_timeline.txt
cat $OUT/${TAG}_precise_bench_under_rocprof.json
