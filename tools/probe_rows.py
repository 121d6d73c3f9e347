"""Times lfd_fasterblock_fused_f16 alone (HIP events) for the library named by LFD_HIP_LIB; with a -DLFD_ROWS_TIMING (and LFD_BLOCK_ROWS=1) build
also dumps the per-step phase stamps of workgroup 0 (producer wave 0 / consumer wave 4)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch  # noqa: E402
from lfd_amd import _lib, ops  # noqa: E402

n, h, w = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 135, 240))]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
g = torch.Generator().manual_seed(0)
w1 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
w2 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
b1, b2 = torch.randn(64, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
x = (torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda()
y = torch.empty_like(x)
import time
t0 = time.time()
while time.time() - t0 < 0.3:          # warm the clocks (see tools/ab_block.py)
    for _ in range(20):
        ops.fasterblock_fused(x, w1, b1, w2, b2, out=y)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.fasterblock_fused(x, w1, b1, w2, b2, out=y)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
rec = dict(lib=os.path.basename(_lib.LIB_PATH), shape=[n, h, w], fused_us=round(us, 2))
l = _lib.lib()
try:
    fn = C.CDLL(_lib.LIB_PATH).lfd_debug_rows_timing
    buf = (C.c_ulonglong * 256)()
    fn(buf)
    import numpy as np
    a = np.array(list(buf), dtype=np.int64).reshape(2, 16, 8)
    t0 = a[0, 0, 0]
    rec['clock_ghz'] = [round(float(a[r, 8, 0] - a[r, 1, 0]) / float(a[r, 8, 7] - a[r, 1, 7]) / 10.0, 3) if a[r, 8, 7] > a[r, 1, 7] else None
                        for r in (0, 1)]      # cycle counter vs the 100 MHz real-time counter over 7 steps
    rec['producer_steps'] = [[int(v - t0) if v else None for v in a[0, s, :6]] for s in range(11)]
    rec['consumer_steps'] = [[int(v - t0) if v else None for v in a[1, s, :6]] for s in range(11)]
    rec['step_wall_ns'] = [[int(a[r, s + 1, 7] - a[r, s, 7]) * 10 if a[r, s + 1, 7] and a[r, s, 7] else None for s in range(10)] for r in (0, 1)]   # 100 MHz real-time counter at each loop top
except AttributeError:
    pass
print(json.dumps(rec))
