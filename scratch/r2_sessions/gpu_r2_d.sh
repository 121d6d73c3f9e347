#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py -q -k rederived --timeout 500 2>&1 | tail -30
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_fullsize.json'))
for k in d:
    if k.startswith('per-launch'):
        for r in d[k].get('rows', []): print(k, r)
PY
