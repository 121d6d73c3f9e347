#!/bin/bash
# round 2, call o: rocprofv3 kernel stats of the training iteration (final kernels), eager and graphed timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 300 python tools/bench_train.py --modes hip,graph 2>&1 | grep -v amdgpu.ids
cd /tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -o tr -- python $R/tools/bench_train.py --modes hip --steps 5 --warmup 2 > /tmp/tr.log 2>&1
find /tmp/p_tr -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02b_train_step_kernel_stats.csv \;
head -12 $R/gpurun_out/r02b_train_step_kernel_stats.csv | cut -c1-150
