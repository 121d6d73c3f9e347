"""Per-phase cycle stamps of the fp32-output conv of the plane head (k_pl_conv<128,1,1,1,..,OUTM 2,PTO 1,GNIN>, csrc/planes_impl.h)
and of the tower conv with GroupNorm on its input, on the first pyramid level of 8 x 1080p (timing build: tools/ab_build.sh
scratch/alt/liblfd_hip_plt.so -DLFD_PL_TIMING; LFD_HIP_LIB=... python tools/timing/pl_out32_phases.py)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import ops, _lib, engine_p2
from lfd_amd._lib import check, lib, ptr, stream_ptr
L = lib()
has_t = hasattr(L, 'lfd_debug_pl_timing')
if has_t: L.lfd_debug_pl_timing.argtypes = [C.c_void_p]
dev = torch.device('cuda')
z = ops.zero_line(dev)
NAMES = ['top-wait', 'barrier', 'dma-issue', 'GN-transform + acc-init + kloop', 'outputs']
def case(tag, n, h, w, cout, out32):
    g = torch.Generator().manual_seed(0)
    xp = engine_p2.to_planes(torch.randn(n, h, w, 128, generator=g)).cuda()
    wt = torch.randn(cout, 128, 1, 1, generator=g) * 0.05
    d = _lib.PlConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = n, h, w, 128, cout, 1, 1, 0
    d.in_plane_halfs = xp[0].numel(); d.gn_in_eps = 1e-5
    wp, b = engine_p2.pack_planes_weight(wt).cuda(), torch.zeros(128, device=dev)
    gsum = torch.zeros((8, n, 16, 2), dtype=torch.int64, device=dev); gsum[0, :, :, 1] = int(h * w * 8 * (1 << 24))
    gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    out = f0 = f1 = gs = None
    if out32:
        d.out_mode, d.f_c0, d.f_c1 = 2, 1, 4
        f0, f1 = torch.empty(n, h * w, 1, device=dev), torch.empty(n, h * w, 4, device=dev)
        d.f_image_stride0, d.f_image_stride1 = h * w, h * w * 4
    else:
        d.out_mode = 1
        out = torch.empty((2, n, h, w, cout), dtype=torch.float16, device=dev); d.out_plane_halfs = out[0].numel()
        gs = torch.zeros((8, n, 16, 2), dtype=torch.int64, device=dev)
    def run():
        check(L.lfd_pl_conv2d(C.byref(d), ptr(xp), ptr(out), ptr(wp), ptr(b), None, None, None, None, None, None, ptr(gs), ptr(f0), ptr(f1), None,
                              ptr(gsum), ptr(gam), ptr(bet), ptr(z), stream_ptr()), tag)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.1f us' % (tag, e0.elapsed_time(e1) * 50))
    if has_t:
        buf = (C.c_ulonglong * 128)(); L.lfd_debug_pl_timing(buf)
        for it in range(1, 4):
            v = [buf[it * 16 + i] for i in range(10)]
            nxt = buf[(it + 1) * 16]
            if out32:
                print('   tile %d: top-wait %d  barrier %d  dma-issue %d  GN-transform + acc-init + kloop %d  outputs.. next top %d' % (it, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], nxt - v[4]))
            else:
                print('   tile %d: ' % it + '  '.join('%d' % (v[i + 1] - v[i]) for i in range(9)) + ' | next-gap %d' % (nxt - v[9]))
case('1x1 128->5 fp32 outputs, GroupNorm on the input, 8x135x240', 8, 135, 240, 5, True)
case('1x1 128->128 + sums, GroupNorm on the input, 8x135x240', 8, 135, 240, 128, False)
