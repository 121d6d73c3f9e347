#!/bin/bash
# same-session A/B of two builds of the library: tools/gpu_ab_bench.sh <alt .so> [rounds]
# (box-to-box differences are 5-20 %: only numbers from one session compare)
cd "$GRAFT_REPO_ROOT" || exit 1
ALT=$PWD/$1; R=${2:-3}
for i in $(seq $R); do
  for tag in cur alt; do
    if [ $tag = alt ]; then export LFD_HIP_LIB=$ALT; else unset LFD_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-train --no-siblings 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['ms_per_step_serial'], d.get('latency_bs1',{}).get('forward_ms',{}).get('p50'))"
  done
done
