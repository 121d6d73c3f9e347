import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import ops, engine
from lfd_amd._lib import check, lib, ptr, stream_ptr
g = torch.Generator().manual_seed(2); c=64
n,h,w = 8,1080,1920
ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5),
      (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5)]
bs = [torch.randn(c, generator=g) * 0.1 for _ in range(4)]
packed = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
bg = [b.cuda() for b in bs]
xin = (torch.rand(n,h,w,3, generator=g)*2-1).half().cuda()
out = torch.empty((n, 270, 480, c), dtype=torch.float16).cuda()
def run():
    check(lib().lfd_stem_faster_fused_f16(ptr(xin), 1, n, h, w, c, ptr(packed[0]), ptr(bg[0]), ptr(packed[1]), ptr(bg[1]),
          ptr(packed[2]), ptr(bg[2]), ptr(packed[3]), ptr(bg[3]), ptr(out), stream_ptr()), 'fused stem')
for _ in range(3): run()
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print('fused stem bs8 1080p: %.1f us (LFD_STEM2X=%s)' % (e0.elapsed_time(e1)*100, os.environ.get('LFD_STEM2X','1')), 'checksum', float(out.float().abs().mean()))
import ctypes
L = lib()
if hasattr(L, 'lfd_debug_x2_timing') and os.environ.get('LFD_STEM2X','1') == '1':
    L.lfd_debug_x2_timing.argtypes=[ctypes.c_void_p]
    buf = (ctypes.c_ulonglong*128)(); L.lfd_debug_x2_timing(buf)
    for it in range(1,4):
        v=[buf[it*16+i] for i in range(10)]
        print('tile',it,'topbar',v[1]-v[0],'phaseA',v[2]-v[1],'barAB',v[3]-v[2],'fetch+xoff',v[4]-v[3],'phaseB',v[5]-v[4],'tail',v[8]-v[5],'waitvm',v[9]-v[8],'copyout',v[6]-v[9],'rawstore',v[7]-v[6],'total',v[7]-v[0])
