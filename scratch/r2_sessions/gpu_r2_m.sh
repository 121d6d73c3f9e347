#!/bin/bash
# round 2, call m: sibling meta-architectures on the device
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_siblings.py -q 2>&1 | tail -30
timeout 600 python tools/bench_siblings.py --no-cpu 2>&1 | grep -v amdgpu.ids | tee gpurun_out/siblings_bench.jsonl
timeout 300 python tools/bench_siblings.py --n 1 --h 1080 --w 1920 --no-cpu 2>&1 | grep -v amdgpu.ids | tee gpurun_out/siblings_bench_bs1.jsonl
