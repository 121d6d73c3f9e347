"""Per-phase cycle stamps of k_pl_conv (csrc/planes_impl.h, built with -DLFD_PL_TIMING: tools/ab_build.sh scratch/alt/liblfd_hip_plt.so
-DLFD_PL_TIMING; LFD_HIP_LIB=... python tools/timing/pl_phases.py) for wave 0 of workgroup 0, first 8 tiles, next to the launch time."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, time
from lfd_amd import ops, _lib, engine_p2
from lfd_amd._lib import check, lib, ptr, stream_ptr
L = lib()
has_t = hasattr(L, 'lfd_debug_pl_timing')
if has_t: L.lfd_debug_pl_timing.argtypes = [C.c_void_p]
dev = torch.device('cuda')
z = ops.zero_line(dev)
NAMES = ['top-wait', 'barrier', 'dma-setup+res', 'acc-init+kloop', 'mid/tail', 'barrier', 'stage-write', 'wait+barrier', 'copy-out']
def case(tag, n, h, w, cin, cout, ks, stride, res=False, tail=False, ds=False, gn=False):
    g = torch.Generator().manual_seed(0)
    xp = engine_p2.to_planes(torch.randn(n, h, w, cin, generator=g)).cuda()
    wt = torch.randn(cout, cin, ks, ks, generator=g) * 0.05
    oh, ow = (h + 2 * (ks // 2) - ks) // stride + 1, (w + 2 * (ks // 2) - ks) // stride + 1
    d = _lib.PlConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = n, h, w, cin, cout, ks, stride, 1
    d.in_plane_halfs = xp[0].numel()
    d.out_mode = 1 if gn else 0
    wp, b = engine_p2.pack_planes_weight(wt).cuda(), torch.zeros(128, device=dev)
    out = torch.empty((2, n, oh, ow, cout), dtype=torch.float16, device=dev); d.out_plane_halfs = out[0].numel()
    rp = tw = dw = dsd = gs = None
    if res: rp = engine_p2.to_planes(torch.randn(n, oh, ow, cout, generator=g)).cuda(); d.res_plane_halfs = rp[0].numel()
    if tail: tw = engine_p2.pack_planes_weight(torch.randn(cout, cout, 1, 1, generator=g) * 0.1).cuda(); d.tail_cout = cout; d.tail_relu = 1
    if ds: dw = engine_p2.pack_planes_weight(torch.randn(cout, cin, 1, 1, generator=g) * 0.1).cuda(); dsd = torch.empty_like(out); d.ds_plane_halfs = out[0].numel()
    if gn: gs = torch.zeros((8, n, 16, 2), dtype=torch.int64, device=dev)
    def run():
        check(L.lfd_pl_conv2d(C.byref(d), ptr(xp), ptr(out), ptr(wp), ptr(b), ptr(rp), ptr(tw), ptr(b) if tail else None, ptr(dw), ptr(b) if ds else None,
                              ptr(dsd), ptr(gs), None, None, None, None, None, None, ptr(z), stream_ptr()), tag)
    t0 = time.time()
    while time.time() - t0 < 0.2: run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.1f us' % (tag, e0.elapsed_time(e1) * 50))
    if has_t:
        buf = (C.c_ulonglong * 128)(); L.lfd_debug_pl_timing(buf)
        for it in range(1, 5):
            v = [buf[it * 16 + i] for i in range(10)]
            print('   tile %d: ' % it + '  '.join('%s %d' % (NAMES[i], v[i + 1] - v[i]) for i in range(9)) + '  | total %d  next-gap %d' % (v[9] - v[0], buf[(it + 1) * 16] - v[9]))
    if hasattr(L, 'lfd_debug_pl_c3_timing') and ks == 3 and stride == 1 and cin == 64 and not (tail or ds or gn):
        L.lfd_debug_pl_c3_timing.argtypes = [C.c_void_p]
        buf = (C.c_ulonglong * 136)(); L.lfd_debug_pl_c3_timing(buf)
        cyc, rt = buf[130] - buf[128], buf[131] - buf[129]
        print('   k_pl_c3 workgroup 0: %d shader cycles in %.2f us -> %.2f GHz' % (cyc, rt / 100.0, cyc / (rt * 10.0)))
        for it in range(1, 4):
            v = [buf[it * 16 + i] for i in range(5)]
            print('   tile %d: top-wait %d barrier %d setup+bias %d kloop %d  next-top %d' % (it, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], buf[(it + 1) * 16] - v[4]))
case('3x3 s1 64 plain 8x135x240', 8, 135, 240, 64, 64, 3, 1)
case('3x3 s1 64 res   8x135x240', 8, 135, 240, 64, 64, 3, 1, res=True)
case('3x3 s2 64 tail  8x540x960', 8, 540, 960, 64, 64, 3, 2, tail=True)
case('3x3 s2 64 ds    8x270x480', 8, 270, 480, 64, 64, 3, 2, ds=True)
case('1x1 64->128 tail gn 8x135x240', 8, 135, 240, 64, 128, 1, 1, tail=True, gn=True)
case('1x1 128->128 gn 8x135x240', 8, 135, 240, 128, 128, 1, 1, gn=True)
case('3x3 s1 64 plain 8x68x120', 8, 68, 120, 64, 64, 3, 1)
# ---- the first stem pair (k_pl_stem)
SN = ['top-barrier+lds-write', 'barrier', 'fetch-issue', 'conv1+mid', 'barrier', 'tail', 'barrier+stage', 'barrier', 'copy-out']
for fmt, n, h, w in [(1, 8, 1080, 1920)]:
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
    print('stem pair fmt %d %dx%dx%d: %.1f us' % (fmt, n, h, w, e0.elapsed_time(e1) * 100))
    if has_t:
        buf = (C.c_ulonglong * 128)(); L.lfd_debug_pl_timing(buf)
        for it in range(1, 5):
            v = [buf[it * 16 + i] for i in range(10)]
            print('   tile %d: ' % it + '  '.join('%s %d' % (SN[i], v[i + 1] - v[i]) for i in range(9)) + '  | total %d  next-gap %d' % (v[9] - v[0], buf[(it + 1) * 16] - v[9]))
