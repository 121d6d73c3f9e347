"""time lfd_pl_stem_pair alone (8 x 1080p fp16 frames) with the library LFD_HIP_LIB points to"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import engine_p2
from lfd_amd._lib import check, lib, ptr, stream_ptr
L = lib(); dev = torch.device('cuda')
n, h, w, fmt = 8, 1080, 1920, 1
g = torch.Generator().manual_seed(0)
x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda()
w1, w2 = torch.randn(64, 3, 3, 3, generator=g) * 0.3, torch.randn(64, 64, 1, 1, generator=g) * 0.1
keep = [engine_p2.pack_planes_stem_weight(w1).cuda(), torch.zeros(64, device=dev), engine_p2.pack_planes_weight(w2).cuda()]
out = torch.empty((2, n, (h + 1) // 2, (w + 1) // 2, 64), dtype=torch.float16, device=dev)
def run():
    check(L.lfd_pl_stem_pair(ptr(x), fmt, n, h, w, 64, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[1]), ptr(out), out[0].numel(), stream_ptr()), 'stem')
for _ in range(5): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print('%s: stem pair %.1f us' % (os.environ.get('LFD_HIP_LIB', 'default'), e0.elapsed_time(e1) * 100))
