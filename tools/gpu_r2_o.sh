#!/bin/bash
# round 2, call o: bench line with the `siblings` key + its contract test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_end2end.py -q -k "bench_prints" 2>&1 | tail -5
