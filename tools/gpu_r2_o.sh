#!/bin/bash
# round 2, call o: A/B of the tile shape of the streamed-weight 128-channel 3x3 conv on large maps (sibling heads) + GN final
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
LFD_CONV128_SPLIT_MINPIX=1 timeout 300 python -m pytest tests/test_gpu_conv.py -q -k "128" 2>&1 | tail -3
LFD_CONV128_SPLIT_MINPIX=1 timeout 300 python -m pytest tests/test_gpu_siblings.py -q -k "forward_on_device or neck_on_device" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_train_convs.py tests/test_gpu_siblings.py -q 2>&1 | tail -3
for v in 1073741824 4096 1073741824 4096; do
  LFD_CONV128_SPLIT_MINPIX=$v timeout 300 python tools/bench_siblings.py --no-cpu --reps 20 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('minpix=$v', d['config'], d['forward_graph_ms']['p50'], d['detect_ms']['p50'])"
done
