#!/bin/bash
# round 2, last call: the default bench line of the final code (inference + latency + train (graph) + siblings + cpu baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 75 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/final_bench.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_serial'], d['roofline'], d['latency_bs1']['end_to_end_ms'], d['train']['ms_per_iter'], d['train']['ms_per_iter_eager'], [(s['config'], s['forward_ms'], s['detect_ms']) for s in d['siblings']], d['cpu_baseline']['value'])
PY
