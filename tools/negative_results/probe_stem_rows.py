"""Eager timing of lfd_stem_faster_fused_f16 (LFD_STEM_ROWS picks the kernel); with a -DLFD_SROWS_TIMING build the phase stamps of
steps 8..23 of workgroup 8 (producer wave 0 / consumer wave 4) and the shader clock."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import _lib, engine, ops
from lfd_amd._lib import check, lib, ptr, stream_ptr
n, h, w = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 1080, 1920))]
g = torch.Generator().manual_seed(0); c = 64
ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / 8), (torch.randn(c, c, 3, 3, generator=g) / 24), (torch.randn(c, c, 1, 1, generator=g) / 8)]
bs = [(torch.randn(c, generator=g) * 0.1).cuda() for _ in range(4)]
pk = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
xs = [(torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda() for _ in range(4)]
y = torch.empty(n, ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2, c, dtype=torch.float16, device='cuda')
def fn(i):
    check(lib().lfd_stem_faster_fused_f16(ptr(xs[i % 4]), 1, n, h, w, c, ptr(pk[0]), ptr(bs[0]), ptr(pk[1]), ptr(bs[1]), ptr(pk[2]), ptr(bs[2]), ptr(pk[3]), ptr(bs[3]), ptr(y), stream_ptr()), 'stem')
t0 = time.time()
while time.time() - t0 < 0.3:
    for i in range(10): fn(i)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40): fn(i)
e1.record(); torch.cuda.synchronize()
rec = dict(lib=os.path.basename(_lib.LIB_PATH), rows=os.environ.get('LFD_STEM_ROWS'), shape=[n, h, w], us=round(e0.elapsed_time(e1) * 1e3 / 40, 2))
try:
    f = C.CDLL(_lib.LIB_PATH).lfd_debug_srows_timing
    buf = (C.c_ulonglong * 256)(); f(buf)
    import numpy as np
    a = np.array(list(buf), dtype=np.int64).reshape(2, 16, 8)
    t0 = a[0, 0, 0]
    rec['clock_ghz'] = round(float(a[0, 12, 0] - a[0, 2, 0]) / float(a[0, 12, 7] - a[0, 2, 7]) / 10.0, 3)
    rec['producer'] = [[int(v - t0) if v else None for v in a[0, s, :3]] for s in range(8)]
    rec['consumer'] = [[int(v - t0) if v else None for v in a[1, s, :6]] for s in range(8)]
except AttributeError:
    pass
print(json.dumps(rec))
