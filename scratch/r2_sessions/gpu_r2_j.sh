#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_block.py tests/test_gpu_forward.py -x -q --timeout 600 2>&1 | tail -3
for n in 1 8; do echo "== breakdown bs $n"; python tools/breakdown_bs1.py $n 2>/dev/null | head -8; done
python bench.py --no-cpu-baseline --no-train 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_serial'], d.get('latency_bs1'))"
