#!/bin/bash
# Same-box A/B of the training iteration's switches (config 5: WIDERFACE_LFD_S 640x640 bs 32, one HIP graph per iteration).
# Every switch is read once per process, so each setting is its own python run.  Usage on the GPU box: tools/ab_train.sh [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}; REPS=${1:-2}
run() {   # label, env assignments...
  local label=$1; shift
  for i in $(seq $REPS); do
    ms=$(env "$@" python $R/tools/bench_train.py --modes graph 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
    echo "$label  $ms ms"
  done
}
run "default                         " LFD_NOOP=1
run "LFD_CONV_BN_STATS=0 (stats pass)" LFD_CONV_BN_STATS=0
run "LFD_DGRAD_S2=0 (zero insert)    " LFD_DGRAD_S2=0
run "default again                   " LFD_NOOP=1
