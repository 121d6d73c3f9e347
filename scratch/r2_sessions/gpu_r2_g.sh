#!/bin/bash
# head statistics per tile / batch-dependent chunk length: parity tests, bs-1 / bs-8 breakdown, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_forward.py -x -q --timeout 900 2>&1 | tail -5
for n in 1 8; do echo "== breakdown bs $n"; python tools/breakdown_bs1.py $n 2>/dev/null | head -8; done
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_g.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_g.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_serial')}, d.get('latency_bs1'))
PY
