#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_end2end.py tests/test_gpu_forward.py tests/test_gpu_postproc.py -q --timeout 500 2>&1 | tail -6
for p in 2 1 3; do timeout 300 python bench.py --pipeline $p --no-cpu-baseline --no-latency --no-train > gpurun_out/e_bench_p$p.json 2>gpurun_out/e_bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/e_bench_p$p.json')); print('pipeline', d['pipeline_depth'], d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'], d['step_ms_hip_events']['median'])
PY
done
