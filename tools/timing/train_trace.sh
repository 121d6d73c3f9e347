#!/bin/bash
# Per-DISPATCH kernel trace of the graph-replayed training iteration: durations in launch order, gaps between consecutive
# kernels, time by duration bucket.  Usage on the GPU box: tools/timing/train_trace.sh TAG [bench_train args] -> gpurun_out/TAG_train_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04}; shift; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/p_tt -o tt -- python $R/tools/bench_train.py --modes graph --steps 6 --warmup 3 "$@" > /tmp/tt.log 2>&1
f=$(find /tmp/p_tt -name '*kernel_trace.csv' | head -1)
m=$(find /tmp/p_tt -name '*memory_copy_trace.csv' | head -1)
python $R/tools/timing/train_trace.py "$f" order "$m" > $OUT/${TAG}_train_trace.txt
head -12 $OUT/${TAG}_train_trace.txt; tail -3 /tmp/tt.log
