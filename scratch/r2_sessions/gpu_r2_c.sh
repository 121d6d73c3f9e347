#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== full gpu suite"; timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -6 | tee gpurun_out/c_full.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
echo "== bench"; timeout 600 python bench.py > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/c_bench.json')); print(d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'], d['step_ms_hip_events'], d['roofline'], d['roofline_conv3x3_s1_64'], d['latency_bs1']['end_to_end_ms'], d['train']['ms_per_iter'], d['cpu_baseline']['value'])
PY
echo "== profiles"; timeout 900 bash tools/collect_profiles.sh r02 2>&1 | tail -4
