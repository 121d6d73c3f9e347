"""Times lfd_downblock_fused_f16 alone (HIP events, eager) for the library named by LFD_HIP_LIB; with a -DLFD_DOWN_TIMING build
also dumps the per-step phase stamps of workgroup 0 (producer wave 0 / consumer wave 4)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch  # noqa: E402
from lfd_amd import _lib, ops  # noqa: E402

n, h, w = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 270, 480))]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
g = torch.Generator().manual_seed(0)
p1 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
pd = ops.pack_conv_weight((torch.randn(64, 64, 1, 1, generator=g) / 8)).cuda()
p2 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
b = torch.randn(64, generator=g).cuda() * 0.1
xs = [(torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda() for _ in range(4)]
y = torch.empty(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 64, dtype=torch.float16, device='cuda')
t0 = time.time()
while time.time() - t0 < 0.3:
    for i in range(20):
        ops.downblock_fused(xs[i % 4], p1, b, pd, b, p2, b, out=y)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps):
    ops.downblock_fused(xs[i % 4], p1, b, pd, b, p2, b, out=y)
e1.record()
torch.cuda.synchronize()
rec = dict(lib=os.path.basename(_lib.LIB_PATH), shape=[n, h, w], fused_us=round(e0.elapsed_time(e1) * 1e3 / reps, 2))
try:
    fn = C.CDLL(_lib.LIB_PATH).lfd_debug_down_timing
    buf = (C.c_ulonglong * 256)()
    fn(buf)
    import numpy as np
    a = np.array(list(buf), dtype=np.int64).reshape(2, 16, 8)
    t0 = a[0, 0, 0]
    rec['clock_ghz'] = round(float(a[0, 10, 0] - a[0, 2, 0]) / float(a[0, 10, 7] - a[0, 2, 7]) / 10.0, 3)
    rec['producer_steps'] = [[int(v - t0) if v else None for v in a[0, s, :6]] for s in range(14)]
    rec['consumer_steps'] = [[int(v - t0) if v else None for v in a[1, s, :6]] for s in range(14)]
except AttributeError:
    pass
print(json.dumps(rec))
