import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch, time
from lfd_amd import ops, _lib
from lfd_amd._lib import check, lib, ptr, stream_ptr
wt = torch.randn(64,64,3,3)*0.05; wp = ops.pack_conv_weight(wt).cuda(); b = torch.zeros(64).cuda()
wd = torch.randn(64,64,1,1)*0.1; wdp = ops.pack_conv_weight(wd).cuda(); bd = torch.zeros(64).cuda()
L = lib()
has_t = hasattr(L, 'lfd_debug_conv_timing')
if has_t: L.lfd_debug_conv_timing.argtypes=[C.c_void_p]
z = ops.zero_line(torch.device('cuda'))
for (n,h,w) in [(8,270,480),(8,135,240),(8,68,120),(1,270,480)]:
    x = torch.randn(n,h,w,64).half().cuda()
    oh, ow = (h+1)//2, (w+1)//2
    out = torch.empty(n,oh,ow,64, dtype=torch.float16, device='cuda'); ods = torch.empty_like(out)
    d = _lib.ConvDesc(n, h, w, 64, 64, 3, 2, 1, 0, 0)
    def run():
        check(L.lfd_conv2d_downsample_nhwc_f16(C.byref(d), ptr(x), ptr(out), ptr(wp), ptr(b), ptr(wdp), ptr(bd), ptr(ods), ptr(z), stream_ptr()), 'ds')
    t0=time.time()
    while time.time()-t0 < 0.3: run(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('s2+ds %dx%dx%d: %.1f us' % (n,h,w, e0.elapsed_time(e1)*50))
    if has_t:
        buf = (C.c_ulonglong*128)(); L.lfd_debug_conv_timing(buf)
        for it in range(3):
            v=[buf[it*16+i] for i in range(9)]
            print('  tile',it,'bar0',v[1]-v[0],'dma issue',v[2]-v[1],'wait+bar',v[3]-v[2],'kloop',v[4]-v[3],'bar2',v[5]-v[4],'stage',v[6]-v[5],'bar3',v[7]-v[6],'copyout',v[8]-v[7],'total',v[8]-v[0], 'next-gap', (buf[(it+1)*16]-v[8]))
