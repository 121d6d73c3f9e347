#!/bin/bash
# round 2, call o: weight-gradient final sum with 16-byte loads: parity + train timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train_convs.py tests/test_gpu_train.py -q -x 2>&1 | tail -4
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-siblings 2>gpurun_out/o_bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['train'])"
