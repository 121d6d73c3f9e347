#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "" "LFD_OVERLAP=1" "LFD_SPLIT_HEAD=1"; do
env $e python - <<'PY' 2>/dev/null | tail -1
import os, sys, json
sys.path.insert(0, '.')
import torch, bench
from lfd_amd import configs
dev = torch.device('cuda', 0)
m = configs.build_model('WIDERFACE_LFD_S'); configs.perturb_weights(m); m.eval().to(dev); m.use_graph = True
m._classification_threshold = 0.745; m._nms_cfg = dict(type='nms', iou_thr=0.4)
with torch.no_grad():
    r = bench.latency_bs1(m, dev)
print(os.environ.get('LFD_OVERLAP'), os.environ.get('LFD_SPLIT_HEAD'), r['forward_ms'], r['end_to_end_ms'])
PY
done
python -m pytest tests/test_gpu_train_convs.py -q -k "end_to_end or whole_network" -s --timeout 300 2>&1 | grep -i "cos\|ratio\|passed\|failed" | head
