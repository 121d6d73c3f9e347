#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q --timeout 600 -k "stem" 2>&1 | tail -3
LFD_HIP_LIB=$PWD/scratch/alt/lib_x2t.so timeout 300 python scratch/stem2x_time.py 2>/dev/null | tail -3
bash tools/gpu_ab_bench.sh scratch/alt/lib_prev.so 2
