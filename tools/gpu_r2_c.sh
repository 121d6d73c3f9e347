#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== full gpu suite"; timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -12 | tee gpurun_out/c_full.log
echo "== bench"; timeout 600 python bench.py > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err; tail -c 600 gpurun_out/c_bench.json
echo "== profiles"; timeout 900 bash tools/collect_profiles.sh r02 2>&1 | tail -40
