#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_end2end.py tests/test_gpu_postproc.py tests/test_gpu_forward.py -x -q --timeout 900 2>&1 | tail -4
for i in 1 2; do for f in 1 0; do LFD_HEAD_DECODE=$f python bench.py --no-cpu-baseline --no-train 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('decode $f', d['value'], d['ms_per_step'], d['ms_per_step_serial'], d['latency_bs1']['forward_ms']['p50'], d['latency_bs1']['end_to_end_ms']['p50'])"; done; done
