#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_end2end.py -x -q --timeout 900 2>&1 | tail -3
for n in 1 8; do echo "== breakdown bs $n"; python tools/breakdown_bs1.py $n 2>/dev/null | grep head; LFD_HEAD_FOLD=0 python tools/breakdown_bs1.py $n 2>/dev/null | grep head; done
for i in 1 2; do for f in 1 0; do LFD_HEAD_FOLD=$f python bench.py --no-cpu-baseline --no-train 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fold $f', d['value'], d['ms_per_step'], d['ms_per_step_serial'], d['latency_bs1']['forward_ms']['p50'])"; done; done
