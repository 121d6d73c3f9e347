#!/bin/bash
# stem: aligned fast path + software-pipelined phase A: parity tests, phase stamps, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q --timeout 600 -k "stem" 2>&1 | tail -5
LFD_HIP_LIB=$PWD/scratch/alt/lib_x2t.so timeout 300 python scratch/stem2x_time.py 2>/dev/null | tail -4
timeout 300 python scratch/stem2x_time.py 2>/dev/null | tail -1
timeout 600 python -m pytest tests/test_gpu_forward.py -x -q --timeout 600 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_h.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_h.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_serial')}, d.get('latency_bs1'))
PY
