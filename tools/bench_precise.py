"""The fp32-storage precision mode alone (LFD.precision = 'fp32_storage') on the headline workload, for rocprofv3:
    rocprofv3 --kernel-trace --stats -- python tools/bench_precise.py [steps]
prints one JSON line (HIP-event median per step, eager launches: under rocprofv3 every launch is a kernel-trace row)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lfd_amd import configs  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = configs.build_model('WIDERFACE_LFD_S')
configs.perturb_weights(m)
m.eval().cuda()
m.precision = 'fp32_storage'
x = (torch.rand(BS, 1080, 1920, 3, device='cuda') * 2 - 1).half()
meta = torch.tensor([[1920., 1080., 1.0]] * BS, device='cuda')
with torch.no_grad():
    cls, _ = m.forward_resident(x)
    m._classification_threshold = float(torch.quantile(cls.float().sigmoid().reshape(BS, -1)[0], 1.0 - 256 / cls.shape[1]))
    for _ in range(3):
        m.detect_resident(x, meta)
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.detect_resident(x, meta)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
print(json.dumps(dict(mode='fp32_storage', workload='WIDERFACE_LFD_S %d x 1920x1080 forward + decode + NMS, eager launches' % BS,
                      ms_per_step=round(float(np.median(ts)), 4), images_per_s=round(BS * 1e3 / float(np.median(ts)), 1), steps=steps)))
