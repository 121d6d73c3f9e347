#!/bin/bash
# round 2, call o: A/B of the neck-filter ring depth in k_head2 (3 vs 5)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 bash tools/gpu_ab_bench.sh scratch/alt/liblfd_hip_w5.so 3
