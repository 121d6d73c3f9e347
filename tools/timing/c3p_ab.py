"""k_pl_c3p (csrc/planes_c3p.hip: the contraction index of the 3x3 s1 64-channel planes conv split over a wave pair per SIMD)
against k_pl_c3 (one wave per SIMD): results vs float64 and launch times, interleaved in one session.
    python tools/timing/c3p_ab.py            (LFD_HIP_LIB=... for a variant build, tools/ab_build.sh)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from lfd_amd import ops, _lib, engine_p2  # noqa: E402
from lfd_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

L = lib()
dev = torch.device('cuda')
z = ops.zero_line(dev)


def make(n, h, w, res, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, 64, generator=g) * 2
    xp = engine_p2.to_planes(x)
    wt = torch.randn(64, 64, 3, 3, generator=g) * (1.0 / 576 ** 0.5)
    b = torch.randn(64, generator=g)
    rp = engine_p2.to_planes(torch.randn(n, h, w, 64, generator=g)) if res else None
    d = _lib.PlConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = n, h, w, 64, 64, 3, 1, 1
    d.in_plane_halfs = xp[0].numel()
    d.out_plane_halfs = xp[0].numel()
    if res:
        d.res_plane_halfs = rp[0].numel()
    keep = dict(xp=xp.cuda(), wp=engine_p2.pack_planes_weight(wt).cuda(), b=engine_p2._pad_bias(b, 128).cuda(), rp=rp.cuda() if res else None,
                out=torch.full((2, n, h, w, 64), float('nan'), dtype=torch.float16, device=dev), d=d, x=engine_p2.from_planes(xp), wt=wt, bias=b,
                r=engine_p2.from_planes(rp) if res else None)
    return keep


def run(k):
    check(L.lfd_pl_conv2d(C.byref(k['d']), ptr(k['xp']), ptr(k['out']), ptr(k['wp']), ptr(k['b']), ptr(k['rp']), None, None, None, None,
                          None, None, None, None, None, None, None, None, ptr(z), stream_ptr()), 'lfd_pl_conv2d')


def ref64(k):
    y = F.conv2d(k['x'].double().permute(0, 3, 1, 2), k['wt'].double(), k['bias'].double(), padding=1).permute(0, 2, 3, 1)
    if k['r'] is not None:
        y = y + k['r'].double()
    return y.relu()


ok = True
for (n, h, w) in [(2, 19, 37), (3, 8, 16), (1, 70, 130), (1, 4, 16), (2, 5, 3)]:
    for res in (False, True):
        k = make(n, h, w, res, seed=h * 7 + w)
        ref = ref64(k)
        errs = {}
        for mode in (1, 2):
            _lib.tune('PL_C3', mode)
            k['out'].fill_(float('nan'))
            run(k); torch.cuda.synchronize()
            got = engine_p2.from_planes(k['out'].cpu()).double()
            errs[mode] = float((got - ref).abs().max()) if not torch.isnan(got).any() else float('nan')
        mag = float(ref.abs().max())
        good = errs[2] == errs[2] and errs[2] <= 4e-6 * max(1.0, mag)
        ok = ok and good
        print('%dx%dx%d res=%d: err c3 %.2e  c3p %.2e  (max |y| %.2f)  %s' % (n, h, w, res, errs[1], errs[2], mag, 'ok' if good else 'FAIL'))
print('CORRECT' if ok else 'WRONG')

# ---- timing, interleaved
t0 = time.time()
kw = make(8, 135, 240, True)
while time.time() - t0 < 0.5:
    run(kw); torch.cuda.synchronize()
rows = []
for (n, h, w) in [(8, 135, 240), (8, 68, 120), (8, 34, 60), (1, 135, 240)]:
    for res in (False, True):
        k = make(n, h, w, res)
        ts = {1: [], 2: []}
        for rep in range(5):
            for mode in (1, 2):
                _lib.tune('PL_C3', mode)
                for _ in range(3):
                    run(k)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(k)
                e1.record(); torch.cuda.synchronize()
                ts[mode].append(e0.elapsed_time(e1) * 50)
        a, b = sorted(ts[1])[2], sorted(ts[2])[2]
        gf = 2 * 3 * n * h * w * 64 * 576 / 1e3
        print('%dx%dx%d res=%d: k_pl_c3 %.1f us (%.0f TF issued)   k_pl_c3p %.1f us (%.0f TF issued)   x%.2f' % (n, h, w, res, a, gf / a / 1e3, b, gf / b / 1e3, a / b))
        rows.append(dict(shape=[n, h, w], res=res, c3_us=round(a, 1), c3p_us=round(b, 1)))
if hasattr(L, 'lfd_debug_pl_c3p_timing'):
    L.lfd_debug_pl_c3p_timing.argtypes = [C.c_void_p]
    for res in (False, True):
        k = make(8, 135, 240, res)
        _lib.tune('PL_C3', 2)
        for _ in range(3):
            run(k)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 136)(); L.lfd_debug_pl_c3p_timing(buf)
        cyc, rt = buf[130] - buf[128], buf[131] - buf[129]
        print('k_pl_c3p res=%d workgroup 0: %d shader cycles in %.2f us -> %.2f GHz' % (res, cyc, rt / 100.0, cyc / (rt * 10.0)))
        for it in range(1, 6):
            v = [buf[it * 16 + i] for i in range(16)]
            print('   tile %d  A: dma-wait %d barrier %d kloop %d handoff %d | B: barrier %d kloop %d tail %d | A period %d  B-top minus A-top %d' % (
                it, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[9] - v[8], v[10] - v[9], buf[(it + 1) * 16 + 8] - v[10], buf[(it + 1) * 16] - v[0], v[8] - v[0]))
print(json.dumps(dict(lib=os.environ.get('LFD_HIP_LIB', 'default'), rows=rows)))
