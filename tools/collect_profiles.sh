#!/bin/bash
# runs on the GPU box from the repo root: kernel-trace stats + HBM traffic counters (separate passes)
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r01; mkdir -p $OUT
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -o kt -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-latency > $OUT/bench_under_rocprof.json 2>/tmp/kt.err
find /tmp/p_kt -name "*kernel_stats.csv" -exec cp {} $OUT/r01_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-latency > /tmp/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python $R/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-latency > /tmp/w.log 2>&1
F=$(find /tmp/p_f -name "*.db" | head -1); W=$(find /tmp/p_w -name "*.db" | head -1)
python $R/tools/pmc_traffic.py $F $W $OUT/r01_pmc_hbm_traffic_raw.json > /tmp/pmc.log 2>&1; tail -3 /tmp/pmc.log
ls -la $OUT
