#!/bin/bash
# round 2, call o: first-conv forward with LDS-staged full-line stores: parity + timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 300 python -m pytest tests/test_gpu_train_convs.py tests/test_gpu_train.py -q -x 2>&1 | tail -3
timeout 300 python tools/bench_train.py --modes graph 2>&1 | grep -v amdgpu.ids
cd /tmp; rm -rf /tmp/p_tr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -o tr -- python $R/tools/bench_train.py --modes hip --steps 5 --warmup 2 > /tmp/tr.log 2>&1
find /tmp/p_tr -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02b_train_step_kernel_stats.csv \;
grep -E "conv0|k_wgrad_final|Name" $R/gpurun_out/r02b_train_step_kernel_stats.csv | cut -c1-170
