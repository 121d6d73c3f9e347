#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_block.py -q -x --timeout 120 2>&1 | tail -2
python tools/probe_ab.py $PWD/scratch/alt/lib_prev.so $PWD/lfd-a-light-and-fast-detector_amd/lfd_amd/liblfd_hip.so 2>/dev/null | tail -4
for v in T; do
LFD_HIP_LIB=$PWD/scratch/alt/lib_$v.so timeout 120 python tools/probe_block.py 8 135 240 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['lib'], d['fused_us'], d.get('clock_ghz'))
for s in (2, 3, 4):
    p, c = d['producer_steps'][s], d['consumer_steps'][s]
    print('  step', s, 'P: bar %d dma %d loop %d epi %d | C: wait %d bar %d loop %d dma %d epi %d | period %d' % (p[2]-p[0], p[3]-p[2], p[4]-p[3], p[5]-p[4], c[1]-c[0], c[2]-c[1], c[4]-c[2], c[3]-c[4], c[5]-c[3], d['producer_steps'][s+1][0]-p[0]))
"
done
