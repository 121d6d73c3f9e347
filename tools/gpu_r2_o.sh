#!/bin/bash
# round 2, call o: k_ex_select with unconditional key loads: parity + detect timing
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 200 python -m pytest tests/test_gpu_siblings.py -q -x -k "detect_ex or topk" 2>&1 | tail -3
timeout 200 python tools/bench_siblings.py --no-cpu --reps 20 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['forward_graph_ms']['p50'], d['detect_ms']['p50'])"
