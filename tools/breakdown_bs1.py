"""per-kernel timing of the forward at batch 1 (1080p): where the bs-1 latency goes"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch, bench
from lfd_amd import configs, engine
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = configs.build_model('WIDERFACE_LFD_S'); configs.perturb_weights(m); m.eval().to(dev)
x = (torch.rand(n, 1080, 1920, 3, device=dev) * 2 - 1).half()
with torch.no_grad():
    m.forward_resident(x)
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3:
        m.forward_resident(x); torch.cuda.synchronize()
    fmt, nn_, h, w = engine._input_format(x)
    plan = engine.get_plan(m, m._backbone, m._neck, m._head, dev)
    st = plan.state_for(nn_, h, w)
    br = bench.kernel_breakdown(m, plan, st, x, fmt, reps=50)
tot = sum(c['time_us'] for c in br.values())
for k, c in sorted(br.items(), key=lambda kv: -kv[1]['time_us']):
    print('%7.1f us  x%d  %s' % (c['time_us'], c['launches'], k))
print('sum %.1f us' % tot)
