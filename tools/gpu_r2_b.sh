#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/b_probe.jsonl
echo "== block tests"; timeout 300 python -m pytest tests/test_gpu_block.py -q -x --timeout 120 2>&1 | tail -3
for v in "" T PRIO; do
  lib=""; [ -n "$v" ] && lib="$PWD/scratch/alt/lib_$v.so"
  for shp in "8 135 240" "32 135 240" "8 68 120"; do
    LFD_HIP_LIB=$lib timeout 120 python tools/probe_block.py $shp 2>/dev/null | tail -1 | cut -c1-1500 | tee -a gpurun_out/b_probe.jsonl
  done
done
timeout 300 python tools/ab_block.py 2>&1 | tail -8
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/b_bench.json')); print(d['ms_per_step'], d['value'], d.get('step_ms_hip_events'), d['latency_bs1']['forward_ms'], d['latency_bs1']['end_to_end_ms'])
for k in d['kernels']: print(k['kernel'][:60], k['launches'], k['time_us_per_forward'], k['frac_mfma'])
PY
