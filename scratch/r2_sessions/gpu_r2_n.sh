#!/bin/bash
# round 2, call n: full GPU suite + smoke after the sibling work (no profiles)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== full gpu suite"; timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -8 | tee gpurun_out/n_full.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
