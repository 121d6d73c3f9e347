#!/bin/bash
# round 2, call m: sibling meta-architectures on the device
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_siblings.py -q 2>&1 | tail -30
timeout 300 python -m pytest tests/test_gpu_train.py tests/test_gpu_forward.py -q -x 2>&1 | tail -4
