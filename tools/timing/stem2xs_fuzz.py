"""Random shapes x three frame formats: k_pl_stem2xs (row stream) against k_pl_stem2x (tiles) through lfd_pl_stem2x -- 90 cases, results
must agree to the order of fp32 sums (round 6: 0 bad, worst 1.9e-6).    python tools/timing/stem2xs_fuzz.py"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch  # noqa: E402
from lfd_amd import ops, _lib, engine_p2
from lfd_amd._lib import check, lib, ptr, stream_ptr
L = lib(); dev = torch.device('cuda'); z = ops.zero_line(dev); c = 64
g = torch.Generator().manual_seed(123)
w1, b1 = torch.randn(c, 3, 3, 3, generator=g) * 0.3, torch.randn(c, generator=g)
w2, b2 = torch.randn(c, c, 1, 1, generator=g) / 8, torch.randn(c, generator=g)
w3, b3 = torch.randn(c, c, 3, 3, generator=g) / 24, torch.randn(c, generator=g)
w4, b4 = torch.randn(c, c, 1, 1, generator=g) / 8, torch.randn(c, generator=g)
keep = [engine_p2.pack_planes_stem2x_weight(w1, b1).cuda(), engine_p2.pack_planes_stem2x_tail_weight(w2).cuda(), engine_p2._pad_bias(b2).cuda(),
        engine_p2.pack_planes_weight(w3).cuda(), engine_p2._pad_bias(b3, 128).cuda(), engine_p2.pack_planes_weight(w4).cuda(), engine_p2._pad_bias(b4, 128).cuda()]
random.seed(7)
worst = 0.0; bad = 0; cases = 0
for it in range(90):
    fmt = it % 3
    mult = {1: 8, 2: 16, 0: 4}[fmt]
    n = random.choice([1, 1, 2, 3, 5, 9])
    h = random.choice([1, 2, 3, 5, 8, 17, 33, 64, 67, 130, 259, 301])
    w = mult * random.choice([1, 2, 3, 5, 9, 17, 33, 40, 61])
    if fmt == 1: x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda()
    elif fmt == 2: x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).cuda()
    else: x = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).cuda()
    oh, ow = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
    outs = []
    for mode in (0, 1):
        _lib.tune('PL_STEM', mode)
        out = torch.full((2, n, oh, ow, c), float('nan'), dtype=torch.float16, device=dev)
        check(L.lfd_pl_stem2x(ptr(x), fmt, n, h, w, *[ptr(k) for k in keep], ptr(out), out[0].numel(), ptr(z), stream_ptr()), 'stem')
        torch.cuda.synchronize()
        outs.append(engine_p2.from_planes(out.cpu()))
    d = (outs[0] - outs[1]).abs()
    nan = int(torch.isnan(outs[1]).sum())
    m = float(d.nan_to_num(1e9).max())
    worst = max(worst, m); cases += 1
    if nan or m > 8e-6:
        bad += 1; print('BAD fmt %d %dx%dx%d: nan %d max diff %.2e' % (fmt, n, h, w, nan, m), flush=True)
print('%d cases, %d bad, worst |tile kernel - stream kernel| = %.2e' % (cases, bad, worst))
_lib.tune('PL_STEM', 1)
