"""Fixed cost per launch of k_pl_c3p: launch time against tiles per workgroup (maps of growing height at 8 x H x 240), events around
100 back-to-back launches.    python tools/timing/c3p_fixed_cost.py          (LFD_HIP_LIB=... for a variant build)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch  # noqa: E402
from lfd_amd import ops, _lib, engine_p2  # noqa: E402
from lfd_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

L = lib()
dev = torch.device('cuda')
z = ops.zero_line(dev)
g = torch.Generator().manual_seed(0)
wt = torch.randn(64, 64, 3, 3, generator=g) / 24
wp = engine_p2.pack_planes_weight(wt).cuda()
b = engine_p2._pad_bias(torch.randn(64, generator=g), 128).cuda()
for (n, h, w) in [(1, 4, 16), (8, 8, 64), (8, 16, 128), (8, 32, 128), (8, 34, 60), (8, 64, 128), (8, 68, 120), (8, 128, 128), (8, 135, 240), (8, 270, 240)]:
    x = (torch.randn(2, n, h, w, 64, device=dev)).half()
    out = torch.empty_like(x)
    d = _lib.PlConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = n, h, w, 64, 64, 3, 1, 1
    d.in_plane_halfs = x[0].numel(); d.out_plane_halfs = x[0].numel()

    def run():
        check(L.lfd_pl_conv2d(C.byref(d), ptr(x), ptr(out), ptr(wp), ptr(b), None, None, None, None, None, None, None, None, None, None,
                              None, None, None, ptr(z), stream_ptr()), 'lfd_pl_conv2d')
    for i in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        run()
    e1.record(); torch.cuda.synchronize()
    tiles = n * ((h + 3) // 4) * ((w + 15) // 16)
    print('%d x %3d x %3d: %5d tiles, %6.2f per workgroup: %6.2f us per launch' % (n, h, w, tiles, tiles / min(256, 8 * ((tiles + 7) // 8)), e0.elapsed_time(e1) * 10), flush=True)
