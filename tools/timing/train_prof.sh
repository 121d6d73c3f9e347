#!/bin/bash
# Per-kernel times of one training iteration (graph replay) + the un-profiled ms/step.  Usage on the GPU box: tools/timing/train_prof.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-train}; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_train.py --modes graph > $OUT/${TAG}_bench_train.json 2>/tmp/bt.err; cat $OUT/${TAG}_bench_train.json
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -o tr -- python $R/tools/bench_train.py --modes graph > /tmp/tr.log 2>&1
f=$(find /tmp/p_tr -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/${TAG}_train_step_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
for r in rows[16:]:
    if 'conv0' in r['Name']:
        print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
