"""The hi/lo-plane kernels of the 'fp32_storage' precision mode (csrc/planes.hip, csrc/planes_impl.h; lfd_amd/engine_p2.py)
against float64 PyTorch of the same op on the values the planes hold:

  * lfd_pl_conv2d: every dispatched (cin, ks, stride, cout) class with each of its options -- chained 1x1 (stem pair,
    neck -> tower conv), second output (a stage's 1x1 stride-2 identity branch, lfd_resnet.py:458-468), residual + ReLU
    (lfd_resnet.py:151-152), GroupNorm sums, fp32 cls / reg outputs at a point offset with Scale (lfd_head.py:176-183,
    lfd.py:526-542) -- on shapes with partial tiles and several tiles per image;
  * lfd_pl_stem_pair: the three frame formats (lfd_resnet.py:356-374; simple_normalize augmentation_pipeline.py:31-36);
  * lfd_pl_groupnorm_relu + the fixed-point sums vs F.group_norm in float64 (lfd_head.py:97-117); sums bit-reproducible;
  * the planes plan == the fp32-tensor plan of round 3 (engine_p32.PrecisePlan) on whole networks to 2e-5.
The whole-network gates against the fp32 oracle (raw <= 1e-4, sigma <= 1e-3) are tests/test_gpu_precise.py: LFD.precision =
'fp32_storage' runs this plan for every configuration it covers.
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from lfd_amd import _lib, configs, engine_p2, engine_p32, ops
from lfd_amd._lib import check, lib, ptr, stream_ptr

pytestmark = pytest.mark.gpu
TOL = 4e-6     # relative to max(1, max |y|): ~22-bit products, fp32 accumulation, one hi/lo split on the way out


def _ref_conv(x, w, b, ks, stride, relu, res=None):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double()
    return y.relu() if relu else y


def _pl_conv(xp, w, b, ks, stride, relu, res=None, tail=None, ds=None, gn=None, out32=None, gnin=None):
    """xp: planes [2,N,H,W,cin] cuda.  Returns planes out (and ds planes) or None for out32."""
    dev = xp.device
    n, h, wd, cin = xp.shape[1:]
    cout = w.shape[0]
    oh, ow = (h + 2 * (ks // 2) - ks) // stride + 1, (wd + 2 * (ks // 2) - ks) // stride + 1
    d = _lib.PlConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = n, h, wd, cin, cout, ks, stride, int(relu)
    d.in_plane_halfs = xp[0].numel()
    keep = [engine_p2.pack_planes_weight(w).to(dev), engine_p2._pad_bias(b, 128).to(dev)]
    out = dsd = None
    tw = tb = dw = db = None
    f0 = f1 = sc = None
    if out32 is not None:
        d.out_mode = 2
        t0, t1, c0, c1, off, P, scale = out32
        d.f_c0, d.f_c1, d.f_image_stride0, d.f_image_stride1 = c0, c1, P * c0, P * c1
        f0 = C.c_void_p(t0.data_ptr() + off * c0 * 4) if c0 else None
        f1 = C.c_void_p(t1.data_ptr() + off * c1 * 4) if c1 else None
        sc = scale
    else:
        d.out_mode = 1 if gn is not None else 0
        out = torch.full((2, n, oh, ow, cout), float('nan'), dtype=torch.float16, device=dev)
        d.out_plane_halfs = out[0].numel()
    if res is not None:
        d.res_plane_halfs = res[0].numel()
    if tail is not None:
        d.tail_cout, d.tail_relu = cout, int(tail[2])
        tw, tb = engine_p2.pack_planes_weight(tail[0]).to(dev), engine_p2._pad_bias(tail[1], 128).to(dev)
    if ds is not None:
        dw, db = engine_p2.pack_planes_weight(ds[0]).to(dev), engine_p2._pad_bias(ds[1], 128).to(dev)
        dsd = torch.full((2, n, oh, ow, cout), float('nan'), dtype=torch.float16, device=dev)
        d.ds_plane_halfs = dsd[0].numel()
    gs = gg = gb = None
    if gnin is not None:
        gs, gg, gb, d.gn_in_eps = gnin
    check(lib().lfd_pl_conv2d(C.byref(d), ptr(xp), ptr(out), ptr(keep[0]), ptr(keep[1]), ptr(res), ptr(tw), ptr(tb), ptr(dw), ptr(db),
                              ptr(dsd), ptr(gn), f0, f1, ptr(sc), ptr(gs), ptr(gg), ptr(gb), ptr(ops.zero_line(dev)), stream_ptr()),
          'lfd_pl_conv2d')
    torch.cuda.synchronize()
    return out, dsd


def _close(got_planes, ref, what):
    got = engine_p2.from_planes(got_planes.cpu()).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert not torch.isnan(got).any(), what + ': unwritten output'
    err, mag = float((got - ref).abs().max()), float(ref.abs().max())
    print('%s: err %.2e (max |y| %.2f)' % (what, err, mag))
    assert err <= TOL * max(1.0, mag), what


def _inputs(seed, n, h, w, cin, cout, ks):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, cin, generator=g) * 2
    xp = engine_p2.to_planes(x)
    xv = engine_p2.from_planes(xp)                       # the value the planes hold (|x - xv| <= 2^-22 |x|)
    wt = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)
    b = torch.randn(cout, generator=g)
    return g, xp.cuda(), xv, wt, b


@pytest.mark.parametrize('cin,cout,ks,stride,n,h,w', [
    (64, 64, 3, 1, 2, 19, 37), (64, 64, 3, 1, 3, 8, 16), (64, 64, 3, 1, 1, 70, 130), (128, 128, 3, 1, 2, 17, 30), (128, 128, 3, 1, 1, 5, 70),
    (64, 64, 3, 2, 2, 33, 50), (64, 128, 3, 2, 2, 34, 60), (128, 128, 3, 2, 2, 21, 39), (32, 32, 3, 2, 2, 20, 28), (32, 64, 3, 2, 1, 41, 77),
    (64, 128, 1, 1, 2, 9, 45), (128, 128, 1, 1, 2, 17, 30), (128, 128, 1, 1, 1, 68, 120)])
def test_pl_conv_plain_and_residual_vs_float64(cin, cout, ks, stride, n, h, w):
    g, xp, xv, wt, b = _inputs(cin * 1000 + cout * 10 + ks + stride + h, n, h, w, cin, cout, ks)
    ref = _ref_conv(xv, wt, b, ks, stride, True)
    got, _ = _pl_conv(xp, wt, b, ks, stride, True)
    _close(got, ref, 'plain %d->%d k%d s%d' % (cin, cout, ks, stride))
    if ks == 3 and stride == 1:
        res = torch.randn(ref.shape, generator=g)
        rp = engine_p2.to_planes(res)
        for relu in (True, False):
            ref2 = _ref_conv(xv, wt, b, ks, stride, relu, engine_p2.from_planes(rp))
            got2, _ = _pl_conv(xp, wt, b, ks, stride, relu, res=rp.cuda())
            _close(got2, ref2, 'residual relu=%d %d->%d' % (relu, cin, cout))


@pytest.mark.parametrize('n,h,w', [(2, 19, 37), (1, 70, 130), (1, 4, 16), (2, 5, 3), (8, 34, 60)])
def test_pl_c3_kernel_variants_vs_float64(n, h, w):
    """The three kernels behind the 3x3 stride-1 64-channel conv (+ residual) of the residual blocks (lfd_resnet.py:96-154),
    selected by LFD_TUNE_PL_C3: 0 generic k_pl_conv, 1 k_pl_c3 (one wave per SIMD), 2 k_pl_c3p (the contraction index split over
    a wave pair per SIMD, round 6: the default) -- each against float64, and against each other (another order of fp32 sums)."""
    g, xp, xv, wt, b = _inputs(64000 + h * 7 + w, n, h, w, 64, 64, 3)
    res = torch.randn(n, h, w, 64, generator=g)
    rp = engine_p2.to_planes(res)
    prev = _lib.tune('PL_C3')
    try:
        for relu, r in ((True, None), (True, rp), (False, rp)):
            ref = _ref_conv(xv, wt, b, 3, 1, relu, None if r is None else engine_p2.from_planes(r))
            got = {}
            for knob in (0, 1, 2):
                _lib.tune('PL_C3', knob)
                out, _ = _pl_conv(xp, wt, b, 3, 1, relu, res=None if r is None else r.cuda())
                _close(out, ref, 'PL_C3=%d relu=%d res=%d' % (knob, relu, r is not None))
                got[knob] = engine_p2.from_planes(out.cpu()).double()
            mag = max(1.0, float(ref.abs().max()))
            assert float((got[2] - got[1]).abs().max()) <= 2e-6 * mag and float((got[2] - got[0]).abs().max()) <= 2e-6 * mag
    finally:
        _lib.tune('PL_C3', prev)


@pytest.mark.parametrize('cin,cout,n,h,w', [(64, 64, 2, 33, 50), (64, 64, 1, 135, 240), (64, 128, 2, 34, 60), (128, 128, 2, 21, 39), (32, 64, 2, 20, 28),
                                            (32, 32, 1, 31, 45)])
def test_pl_conv_stride2_with_identity_branch_vs_float64(cin, cout, n, h, w):
    """first block of a stage (lfd_resnet.py:458-468): conv3x3 s2 + BN + ReLU and the 1x1 s2 + BN identity branch in one launch"""
    g, xp, xv, wt, b = _inputs(7 + cin + cout + h, n, h, w, cin, cout, 3)
    wd_, bd_ = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5), torch.randn(cout, generator=g)
    got, dsd = _pl_conv(xp, wt, b, 3, 2, True, ds=(wd_, bd_))
    _close(got, _ref_conv(xv, wt, b, 3, 2, True), 'entry main %d->%d' % (cin, cout))
    _close(dsd, _ref_conv(xv, wd_, bd_, 1, 2, False), 'entry identity %d->%d' % (cin, cout))


@pytest.mark.parametrize('cin,c,ks,stride,n,h,w', [(64, 64, 3, 2, 2, 33, 50), (64, 64, 3, 2, 1, 100, 180), (32, 32, 3, 2, 2, 20, 28),
                                                   (64, 128, 1, 1, 2, 9, 45), (128, 128, 1, 1, 2, 17, 30), (64, 128, 1, 1, 1, 68, 120)])
def test_pl_conv_chained_1x1_vs_float64(cin, c, ks, stride, n, h, w):
    """stem pair conv3x3 s2 + ReLU -> conv1x1 + ReLU (lfd_resnet.py:396-413); neck conv + ReLU -> first tower conv (no ReLU:
    GroupNorm follows) with the GroupNorm sums of the stored values"""
    g, xp, xv, wt, b = _inputs(11 + cin + c + h, n, h, w, cin, c, ks)
    w2, b2 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    relu2 = ks == 3
    mid = _ref_conv(xv, wt, b, ks, stride, True)
    # the intermediate is split into planes on its way to the second contraction: reference on the same values
    midv = engine_p2.from_planes(engine_p2.to_planes(mid.float())).double()
    ref = _ref_conv(midv, w2, b2, 1, 1, relu2)
    gn = torch.zeros((_lib.PL_GN_REPLICAS, n, c // 8, 2), dtype=torch.int64, device='cuda') if ks == 1 else None
    got, _ = _pl_conv(xp, wt, b, ks, stride, True, tail=(w2, b2, relu2), gn=gn)
    err_mid = float((mid - midv).abs().max())
    _close(got, ref, 'chained %d->%d->%d k%d (split of the intermediate %.1e)' % (cin, c, c, ks, err_mid))
    if gn is not None:
        _check_sums(gn, got)


def _check_sums(gn, planes):
    v = engine_p2.from_planes(planes.cpu()).double()            # [n, h, w, c]
    n, c = v.shape[0], v.shape[3]
    vg = v.reshape(n, -1, c // 8, 8)
    s, q = vg.sum((1, 3)), (vg * vg).sum((1, 3))
    sabs = float(vg.abs().sum((1, 3)).max())
    got = gn.cpu().sum(0).double() / 2.0 ** 24          # (the replicas a launch spreads its atomics over)
    assert float((got[..., 0] - s).abs().max()) <= 2e-6 * sabs + 1e-5
    assert float((got[..., 1] - q).abs().max()) <= 2e-6 * float(q.abs().max()) + 1e-5


@pytest.mark.parametrize('n,h,w', [(2, 17, 30), (1, 135, 240), (3, 1, 5)])
def test_pl_tower_conv_sums_and_groupnorm_vs_float64(n, h, w):
    """conv1x1 128 -> 128 with GroupNorm sums, then GroupNorm(16, 128) + ReLU in place (lfd_head.py:97-117)"""
    g, xp, xv, wt, b = _inputs(3 + h, n, h, w, 128, 128, 1)
    gamma, beta = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    sums = torch.zeros((_lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device='cuda')
    got, _ = _pl_conv(xp, wt, b, 1, 1, False, gn=sums)
    _close(got, _ref_conv(xv, wt, b, 1, 1, False), 'tower conv')
    _check_sums(sums, got)
    # bit-reproducible statistics: order-independent fixed-point atomics
    sums2 = torch.zeros_like(sums)
    got2, _ = _pl_conv(xp, wt, b, 1, 1, False, gn=sums2)
    assert torch.equal(sums, sums2) and torch.equal(got, got2)
    yv = engine_p2.from_planes(got.cpu()).double()
    ref = F.group_norm(yv.reshape(n, h * w, 128).permute(0, 2, 1), 16, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1).relu()
    gd, bd = gamma.cuda(), beta.cuda()
    pre = got.clone()
    check(lib().lfd_pl_groupnorm_relu(ptr(got), got[0].numel(), n, h * w, 128, ptr(sums), ptr(gd), ptr(bd), 1e-5, 1, stream_ptr()),
          'lfd_pl_groupnorm_relu')
    torch.cuda.synchronize()
    out = engine_p2.from_planes(got.cpu()).double().reshape(n, h * w, 128)
    assert float((out - ref).abs().max()) <= 6e-6 * max(1.0, float(ref.abs().max()))
    # the same normalisation applied by the CONSUMER to the tile it fetched: tower conv (with its own sums) and output convs
    refv = engine_p2.from_planes(engine_p2.to_planes(ref.float().reshape(n, h, w, 128))).double()   # (what the kernel's LDS tile holds)
    w2, b2 = torch.randn(128, 128, 1, 1, generator=g) * 0.09, torch.randn(128, generator=g)
    sums2 = torch.zeros_like(sums)
    got2, _ = _pl_conv(pre, w2, b2, 1, 1, False, gn=sums2, gnin=(sums, gd, bd, 1e-5))
    _close(got2, _ref_conv(refv, w2, b2, 1, 1, False), 'tower conv on the normalised tile')
    _check_sums(sums2, got2)
    wo, bo = torch.randn(5, 128, 1, 1, generator=g) * 0.1, torch.randn(5, generator=g)
    cls, reg = torch.zeros((n, h * w, 1), device='cuda'), torch.zeros((n, h * w, 4), device='cuda')
    _pl_conv(pre, wo, bo, 1, 1, False, out32=(cls, reg, 1, 4, 0, h * w, None), gnin=(sums, gd, bd, 1e-5))
    ro = _ref_conv(refv, wo, bo, 1, 1, False).reshape(n, h * w, 5)
    err = max(float((cls.cpu().double() - ro[..., :1]).abs().max()), float((reg.cpu().double() - ro[..., 1:]).abs().max()))
    assert err <= 8e-6 * max(1.0, float(ro.abs().max())), err


@pytest.mark.parametrize('ccls,merged', [(1, True), (46, False), (3, True)])
def test_pl_output_convs_write_fp32_into_the_level_concatenated_tensors(ccls, merged):
    """cls + reg convs (+ Scale on reg) straight into [N,P,C'] / [N,P,4] at a point offset (lfd.py:526-542); everything outside
    the level's rows stays untouched"""
    g, xp, xv, _, _ = _inputs(ccls, 3, 5, 37, 128, 32, 1)
    P, off, hw = 300, 41, 5 * 37
    wc, bc = torch.randn(ccls, 128, 1, 1, generator=g) * 0.1, torch.randn(ccls, generator=g)
    wr, br = torch.randn(4, 128, 1, 1, generator=g) * 0.1, torch.randn(4, generator=g)
    sc = torch.tensor([1.37])
    cls = torch.full((3, P, ccls), -7.0, device='cuda')
    reg = torch.full((3, P, 4), -7.0, device='cuda')
    scd = sc.cuda()
    if merged:
        _pl_conv(xp, torch.cat([wc, wr]), torch.cat([bc, br]), 1, 1, False, out32=(cls, reg, ccls, 4, off, P, scd))
    else:
        _pl_conv(xp, wc, bc, 1, 1, False, out32=(cls, None, ccls, 0, off, P, None))
        _pl_conv(xp, wr, br, 1, 1, False, out32=(None, reg, 0, 4, off, P, scd))
    rc = _ref_conv(xv, wc, bc, 1, 1, False).reshape(3, hw, ccls)
    rr = (_ref_conv(xv, wr, br, 1, 1, False) * 1.37).reshape(3, hw, 4)
    c, r = cls.cpu(), reg.cpu()
    assert float((c[:, off:off + hw].double() - rc).abs().max()) < 1e-5
    assert float((r[:, off:off + hw].double() - rr).abs().max()) < 1e-5
    for t in (c, r):
        assert bool((t[:, :off] == -7).all()) and bool((t[:, off + hw:] == -7).all())


@pytest.mark.parametrize('fmt', [0, 1, 2])
@pytest.mark.parametrize('c', [64, 32])
def test_pl_stem_pair_vs_float64(fmt, c):
    g = torch.Generator().manual_seed(fmt * 7 + c)
    n, h, w = 2, 37, 131
    w1, b1 = torch.randn(c, 3, 3, 3, generator=g) * 0.3, torch.randn(c, generator=g)
    w2, b2 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    if fmt == 0:
        x = torch.rand(n, 3, h, w, generator=g) * 2 - 1
        xr = x.permute(0, 2, 3, 1)
    elif fmt == 1:
        x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half()
        xr = x.float()
    else:
        x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
        xr = (x.float() / 255 - 0.5) / 0.5
    xr = engine_p2.from_planes(engine_p2.to_planes(xr))
    mid = _ref_conv(xr, w1, b1, 3, 2, True)
    ref = _ref_conv(engine_p2.from_planes(engine_p2.to_planes(mid.float())).double(), w2, b2, 1, 1, True)
    oh, ow = (h + 1) // 2, (w + 1) // 2
    out = torch.full((2, n, oh, ow, c), float('nan'), dtype=torch.float16, device='cuda')
    xd = x.cuda()
    keep = [engine_p2.pack_planes_stem_weight(w1).cuda(), engine_p2._pad_bias(b1).cuda(), engine_p2.pack_planes_weight(w2).cuda(),
            engine_p2._pad_bias(b2).cuda()]
    check(lib().lfd_pl_stem_pair(ptr(xd), fmt, n, h, w, c, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(out), out[0].numel(),
                                 stream_ptr()), 'lfd_pl_stem_pair')
    torch.cuda.synchronize()
    _close(out, ref, 'stem pair fmt %d c %d' % (fmt, c))


@pytest.mark.parametrize('fmt,n,h,w', [(1, 2, 75, 132), (1, 1, 270, 480), (1, 2, 37, 131), (0, 2, 37, 131), (2, 1, 70, 133), (1, 1, 8, 6),
                                       (1, 3, 129, 258), (1, 2, 75, 136), (1, 2, 37, 128), (1, 1, 8, 8), (1, 3, 129, 264), (1, 1, 16, 2000),
                                       (1, 2, 1080, 64), (1, 1, 4, 8), (1, 5, 2, 16), (2, 2, 75, 144), (2, 1, 270, 480), (2, 3, 129, 272),
                                       (2, 1, 8, 16), (2, 1, 16, 2000), (2, 4, 3, 32), (0, 2, 75, 136), (0, 1, 270, 480), (0, 3, 129, 260),
                                       (0, 1, 8, 8), (0, 4, 3, 32), (0, 1, 16, 2000)])
@pytest.mark.parametrize('stem_kernel', [1, 0])
def test_pl_stem2x_vs_float64(fmt, n, h, w, stem_kernel):
    """the whole 'faster' stem in one launch (lfd_pl_stem2x) against float64 convs on the values the planes hold
    (lfd_resnet.py:376-413); fp16 / uint8 NHWC and fp32 NCHW frames with 16-byte aligned rows run the row-stream kernel k_pl_stem2xs (tuning knob
    PL_STEM = 1, the default) or the tile kernel k_pl_stem2x (0); the rest the tile kernel's load path"""
    if stem_kernel == 0 and not ((fmt == 1 and w % 8 == 0) or (fmt == 2 and w % 16 == 0) or (fmt == 0 and w % 4 == 0)):
        pytest.skip('same kernel as PL_STEM = 1 for this format')
    _lib.tune('PL_STEM', stem_kernel)
    try:
        _stem2x_case(fmt, n, h, w)
    finally:
        _lib.tune('PL_STEM', 1)


def _stem2x_case(fmt, n, h, w):
    g = torch.Generator().manual_seed(fmt * 11 + h)
    c = 64
    w1, b1 = torch.randn(c, 3, 3, 3, generator=g) * 0.3, torch.randn(c, generator=g)
    w2, b2 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    w3, b3 = torch.randn(c, c, 3, 3, generator=g) * (1.0 / (9 * c) ** 0.5), torch.randn(c, generator=g)
    w4, b4 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    if fmt == 0:
        x = torch.rand(n, 3, h, w, generator=g) * 2 - 1
        xr = x.permute(0, 2, 3, 1)
    elif fmt == 1:
        x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half()
        xr = x.float()
    else:
        x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
        xr = (x.float() / 255 - 0.5) / 0.5

    def rt(t):        # round trip through planes: what the next conv is fed
        return engine_p2.from_planes(engine_p2.to_planes(t.float())).double()
    y = _ref_conv(rt(xr), w1, b1, 3, 2, True)
    y = _ref_conv(rt(y), w2, b2, 1, 1, True)
    y = _ref_conv(rt(y), w3, b3, 3, 2, True)
    ref = _ref_conv(rt(y), w4, b4, 1, 1, True)
    oh, ow = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
    out = torch.full((2, n, oh, ow, c), float('nan'), dtype=torch.float16, device='cuda')
    xd = x.cuda()
    keep = [engine_p2.pack_planes_stem2x_weight(w1, b1).cuda(), engine_p2.pack_planes_stem2x_tail_weight(w2).cuda(),
            engine_p2._pad_bias(b2).cuda(), engine_p2.pack_planes_weight(w3).cuda(), engine_p2._pad_bias(b3, 128).cuda(),
            engine_p2.pack_planes_weight(w4).cuda(), engine_p2._pad_bias(b4, 128).cuda()]
    check(lib().lfd_pl_stem2x(ptr(xd), fmt, n, h, w, *[ptr(k) for k in keep], ptr(out), out[0].numel(), ptr(ops.zero_line(xd.device)),
                              stream_ptr()), 'lfd_pl_stem2x')
    torch.cuda.synchronize()
    _close(out, ref, 'stem2x fmt %d %dx%dx%d' % (fmt, n, h, w))


@pytest.mark.parametrize('fmt', [1, 2, 0])
def test_pl_stem2x_stream_kernel_is_deterministic_under_concurrent_work(fmt):
    """k_pl_stem2xs hands chunks from producer waves to consumer waves through LDS with ONE workgroup barrier per slot (no atomics, fixed
    order): the same frames must give the same bits on every launch, also with another stream keeping the chip busy -- an LDS hazard
    would show here as a bit that differs"""
    n, h, w, c = 3, 258, 528, 64
    g = torch.Generator().manual_seed(77)
    w1, b1 = torch.randn(c, 3, 3, 3, generator=g) * 0.3, torch.randn(c, generator=g)
    w2, b2 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    w3, b3 = torch.randn(c, c, 3, 3, generator=g) * (1.0 / (9 * c) ** 0.5), torch.randn(c, generator=g)
    w4, b4 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    if fmt == 1:
        x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda()
    elif fmt == 2:
        x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).cuda()
    else:
        x = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).cuda()
    keep = [engine_p2.pack_planes_stem2x_weight(w1, b1).cuda(), engine_p2.pack_planes_stem2x_tail_weight(w2).cuda(),
            engine_p2._pad_bias(b2).cuda(), engine_p2.pack_planes_weight(w3).cuda(), engine_p2._pad_bias(b3, 128).cuda(),
            engine_p2.pack_planes_weight(w4).cuda(), engine_p2._pad_bias(b4, 128).cuda()]
    oh, ow = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
    out = torch.empty((2, n, oh, ow, c), dtype=torch.float16, device='cuda')
    z = ops.zero_line(x.device)
    assert _lib.tune('PL_STEM') == 1

    def run():
        out.fill_(float('nan'))
        check(lib().lfd_pl_stem2x(ptr(x), fmt, n, h, w, *[ptr(k) for k in keep], ptr(out), out[0].numel(), ptr(z), stream_ptr()), 'lfd_pl_stem2x')
        torch.cuda.synchronize()
    run()
    ref = out.clone()
    assert not torch.isnan(ref.float()).any()
    side, junk = torch.cuda.Stream(), torch.rand(2048, 2048, device='cuda')
    for i in range(40):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                junk = (junk @ junk).clamp_(-1, 1)
        run()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), 'launch %d differs from the first one' % i
    torch.cuda.synchronize()


def test_pl_conv2d_levels_equals_one_launch_per_level_and_validates_its_arguments():
    """lfd_pl_conv2d_levels at the C ABI: three feature maps of different sizes with their own filters in one launch ==
    lfd_pl_conv2d per level, bit for bit (planes and the fixed-point GroupNorm sums); one level; argument checks"""
    n = 2
    shapes = [(23, 41), (12, 21), (1, 3)]
    g = torch.Generator().manual_seed(17)
    per_level, keep = [], []
    arr = (_lib.PlLevel * len(shapes))()
    for j, (h, w) in enumerate(shapes):
        _, xp, xv, wt, b = _inputs(40 + j, n, h, w, 128, 128, 1)
        sums = torch.zeros((_lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device='cuda')
        ref_out, _ = _pl_conv(xp, wt, b, 1, 1, False, gn=sums)
        out = torch.full_like(ref_out, float('nan'))
        sums_ml = torch.zeros_like(sums)
        wp, bp = engine_p2.pack_planes_weight(wt).cuda(), engine_p2._pad_bias(b, 128).cuda()
        keep += [xp, wp, bp, out, sums_ml]
        lv = arr[j]
        lv.in_, lv.out, lv.w_packed, lv.bias, lv.gn_sums = xp.data_ptr(), out.data_ptr(), wp.data_ptr(), bp.data_ptr(), sums_ml.data_ptr()
        lv.h, lv.w, lv.in_plane_halfs, lv.out_plane_halfs = h, w, xp[0].numel(), out[0].numel()
        per_level.append((ref_out, sums, out, sums_ml))
    d = _lib.PlConvDesc()
    d.n, d.cin, d.cout, d.ks, d.stride, d.relu, d.out_mode = n, 128, 128, 1, 1, 0, 1
    z = ops.zero_line(torch.device('cuda'))
    check(lib().lfd_pl_conv2d_levels(C.byref(d), arr, len(shapes), ptr(z), stream_ptr()), 'lfd_pl_conv2d_levels')
    torch.cuda.synchronize()
    for ref_out, sums, out, sums_ml in per_level:      # (which replica a workgroup adds to depends on the grid: their SUM is the statistic)
        assert torch.equal(out, ref_out) and torch.equal(sums_ml.sum(0), sums.sum(0))
    # a single level is a valid call
    per_level[0][2].fill_(float('nan'))
    per_level[0][3].zero_()
    check(lib().lfd_pl_conv2d_levels(C.byref(d), arr, 1, ptr(z), stream_ptr()), 'lfd_pl_conv2d_levels')
    torch.cuda.synchronize()
    assert torch.equal(per_level[0][2], per_level[0][0]) and torch.equal(per_level[0][3].sum(0), per_level[0][1].sum(0))
    # argument checks: LFD_ERR_INVALID_ARGUMENT (-1) / LFD_ERR_UNSUPPORTED (-4), nothing launched
    assert lib().lfd_pl_conv2d_levels(C.byref(d), arr, 0, ptr(z), stream_ptr()) == -1
    assert lib().lfd_pl_conv2d_levels(C.byref(d), arr, _lib.MAX_LEVELS + 1, ptr(z), stream_ptr()) == -1
    assert lib().lfd_pl_conv2d_levels(C.byref(d), None, 1, ptr(z), stream_ptr()) == -1
    d.ks = 3
    assert lib().lfd_pl_conv2d_levels(C.byref(d), arr, len(shapes), ptr(z), stream_ptr()) == -4
    d.ks = 1
    saved = arr[1].gn_sums
    arr[1].gn_sums = None                      # out_mode 1 without sums
    assert lib().lfd_pl_conv2d_levels(C.byref(d), arr, len(shapes), ptr(z), stream_ptr()) == -1
    arr[1].gn_sums = saved
    d.cout = 96                                # no instance
    assert lib().lfd_pl_conv2d_levels(C.byref(d), arr, len(shapes), ptr(z), stream_ptr()) == -4


def test_pl_stem2x_validates_its_arguments():
    x = torch.zeros((1, 16, 16, 3), dtype=torch.float16, device='cuda')
    w = torch.zeros(1 << 17, dtype=torch.float16, device='cuda')       # >= the largest packed filter (3x3 64 -> 64: 73,728 halfs)
    b = torch.zeros(128, device='cuda')
    out = torch.zeros((2, 1, 4, 4, 64), dtype=torch.float16, device='cuda')
    z = ops.zero_line(x.device)
    L = lib()
    args = lambda fmt=1, xin=x, o=out, plane=None: (ptr(xin), fmt, 1, 16, 16, ptr(w), ptr(w), ptr(b), ptr(w), ptr(b), ptr(w), ptr(b), ptr(o),   # noqa: E731
                                                     o[0].numel() if plane is None else plane, ptr(z), stream_ptr())
    assert L.lfd_pl_stem2x(*args()) == 0
    assert L.lfd_pl_stem2x(*args(fmt=7)) == -1
    assert L.lfd_pl_stem2x(*args(plane=out[0].numel() + 4)) == -1          # plane stride not a multiple of 8 halfs
    assert L.lfd_pl_stem2x(None, 1, 1, 16, 16, ptr(w), ptr(w), ptr(b), ptr(w), ptr(b), ptr(w), ptr(b), ptr(out), out[0].numel(), ptr(z),
                           stream_ptr()) == -1
    torch.cuda.synchronize()


def test_pl_conv_refuses_what_it_has_no_instance_for():
    xp = torch.zeros((2, 1, 8, 8, 64), dtype=torch.float16, device='cuda')
    d = _lib.PlConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = 1, 8, 8, 64, 96, 3, 1, 1
    d.in_plane_halfs = d.out_plane_halfs = xp[0].numel()
    w, b = torch.zeros(1 << 16, dtype=torch.float16, device='cuda'), torch.zeros(128, device='cuda')
    out = torch.zeros((2, 1, 8, 8, 96), dtype=torch.float16, device='cuda')
    z = ops.zero_line(xp.device)
    rc = lib().lfd_pl_conv2d(C.byref(d), ptr(xp), ptr(out), ptr(w), ptr(b), None, None, None, None, None, None, None, None, None, None,
                             None, None, None, ptr(z), stream_ptr())
    assert rc == -4       # LFD_ERR_UNSUPPORTED
    d.cout = 64
    rc = lib().lfd_pl_conv2d(C.byref(d), ptr(xp), None, ptr(w), ptr(b), None, None, None, None, None, None, None, None, None, None,
                             None, None, None, ptr(z), stream_ptr())
    assert rc == -1       # out == NULL
    # a chained 1x1 behind a conv whose output channels do not fit ONE workgroup's slabs (64 -> 128 3x3 s2 + tail on the
    # <64,3,2,2> instance: two cout groups, but the chained operand needs all 128 channels in one): refused, not miscomputed
    d.cout, d.stride, d.tail_cout, d.tail_relu = 128, 2, 128, 1
    out2 = torch.zeros((2, 1, 4, 4, 128), dtype=torch.float16, device='cuda')
    d.out_plane_halfs = out2[0].numel()
    rc = lib().lfd_pl_conv2d(C.byref(d), ptr(xp), ptr(out2), ptr(w), ptr(b), None, ptr(w), ptr(b), None, None, None, None, None, None, None,
                             None, None, None, ptr(z), stream_ptr())
    assert rc == -4


@pytest.mark.parametrize('name,shape', [('WIDERFACE_LFD_S', (2, 135, 241)), ('WIDERFACE_LFD_XS', (1, 96, 128)), ('TT100K_LFD_L', (2, 90, 161)),
                                        ('WIDERFACE_LFD_L', (1, 100, 156))])
def test_planes_plan_equals_the_fp32_tensor_plan(name, shape):
    """the two forms of the mode (hi/lo planes, csrc/planes.hip; fp32 tensors, csrc/precise.hip) agree to 2e-5 on the raw logits"""
    m = configs.build_model(name)
    configs.perturb_weights(m)
    m.eval().cuda()
    m.precision = 'fp32_storage'
    x = (torch.rand(shape[0], 3, shape[1], shape[2], generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
    from lfd_amd import engine_p32 as e32
    with torch.no_grad():
        plan = e32.get_plan(m, x.device)
        assert isinstance(plan, engine_p2.PlanesPlan)
        c, r = m(x)
        c2, r2 = m(x)
        assert torch.equal(c, c2) and torch.equal(r, r2)          # deterministic (fixed-point GroupNorm sums)
        c, r = c.clone(), r.clone()
        # the neck / head convs as one launch over all pyramid levels (lfd_pl_conv2d_levels, the default) == one launch per
        # level (lfd_pl_conv2d), bit for bit
        assert plan.level_groups is not None and len(plan.level_groups) < len(plan.ops) - plan.head_start
        # round 6: heads of the merged-path form run the flat-tile kernels (lfd_pl_head_levels): another order of sums -- close
        # to, not bit-equal with, the generic conv; LFD_P2_FLATHEAD=0 selects the generic multi-level conv, which IS bit-equal
        # to one launch per level
        flat = plan._flat_head_modes() is not None
        os.environ['LFD_P2_FLATHEAD'] = '0'
        try:
            cg, rg = [t.clone() for t in m(x)]
            os.environ['LFD_P2_LEVELS'] = '0'
            try:
                c1, r1 = m(x)
                assert torch.equal(cg, c1) and torch.equal(rg, r1)
            finally:
                del os.environ['LFD_P2_LEVELS']
        finally:
            del os.environ['LFD_P2_FLATHEAD']
        eg = max(float((c - cg).abs().max()), float((r - rg).abs().max()))
        print('%s: flat-tile head %s; vs the generic multi-level conv %.2e' % (name, flat, eg))
        assert (eg <= 2e-5) if flat else (eg == 0.0)
        if plan.stem2x is not None:
            # the whole 'faster' stem as one launch (lfd_pl_stem2x; the default for resident fp16 frames) on this fp32 input
            # == the two-launch stem up to the summation order of conv0
            os.environ['LFD_P2_STEM2X'] = '2'
            try:
                cf, rf = m(x)
                ef = max(float((c - cf).abs().max()), float((r - rf).abs().max()))
                print('%s: one-launch stem vs two launches %.2e' % (name, ef))
                assert ef <= 2e-5
            finally:
                del os.environ['LFD_P2_STEM2X']
        old = os.environ.get('LFD_P32_PLANES')
        os.environ['LFD_P32_PLANES'] = '0'
        try:
            assert isinstance(e32.get_plan(m, x.device), e32.PrecisePlan)
            c0, r0 = m(x)
        finally:
            if old is None:
                del os.environ['LFD_P32_PLANES']
            else:
                os.environ['LFD_P32_PLANES'] = old
    err = max(float((c - c0).abs().max()), float((r - r0).abs().max()))
    print('%s: planes vs fp32 tensors %.2e' % (name, err))
    assert err <= 2e-5


@pytest.mark.parametrize('cin,sizes,n,ccls', [(64, [(19, 37), (9, 15), (5, 14)], 2, 1), (128, [(17, 30), (3, 5)], 3, 1), (64, [(8, 8)], 1, 46)])
def test_pl_head_levels_chain_vs_float64(cin, sizes, n, ccls):
    """lfd_pl_head_levels (csrc/planes_head.hip, round 6): the neck + head of lfd_head.py:164-185 over flat pixel tiles --
    mode 0 (tap planes -> neck 1x1 + ReLU -> first tower 1x1, fp32 out + GroupNorm sums), mode 1 (GroupNorm + ReLU -> tower
    1x1), mode 2 (GroupNorm + ReLU -> cls | reg 1x1 (+ Scale) into the level-concatenated fp32 outputs) -- several pyramid
    levels per launch with their own filters, partial last tiles, against the same chain in float64."""
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(cin + n + len(sizes))
    z = ops.zero_line(dev)
    P = [h * w for h, w in sizes]
    Ptot, poff = sum(P), [sum(P[:i]) for i in range(len(P))]
    cls = torch.full((n, Ptot, ccls), float('nan'), device=dev)
    reg = torch.full((n, Ptot, 4), float('nan'), device=dev)
    gsum = torch.zeros((2, _lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device=dev)
    keep, lv0, lv1, lv2, refs, w3s = [], (_lib.PlHeadLevel * len(P))(), (_lib.PlHeadLevel * len(P))(), (_lib.PlHeadLevel * len(P))(), [], []
    for i, p in enumerate(P):
        x = torch.randn(n, p, cin, generator=g) * 2
        xp = engine_p2.to_planes(x.reshape(n, p, 1, cin))
        xv = engine_p2.from_planes(xp).reshape(n, p, cin).double()
        w0, b0 = torch.randn(128, cin, generator=g) / cin ** 0.5, torch.randn(128, generator=g)
        w1, b1 = torch.randn(128, 128, generator=g) / 128 ** 0.5, torch.randn(128, generator=g)
        w2, b2 = torch.randn(128, 128, generator=g) / 128 ** 0.5, torch.randn(128, generator=g)
        w3, b3 = torch.randn(ccls + 4, 128, generator=g) / 128 ** 0.5, torch.randn(ccls + 4, generator=g)
        ga1, be1, ga2, be2 = [torch.randn(128, generator=g) * s + o for s, o in ((0.3, 1.0), (0.3, 0.0), (0.3, 1.0), (0.3, 0.0))]
        scale = torch.tensor([1.7 + i])
        t1 = (xv @ w0.double().t() + b0.double()).relu() @ w1.double().t() + b1.double()
        y1 = F.group_norm(t1.permute(0, 2, 1), 16, ga1.double(), be1.double(), 1e-5).permute(0, 2, 1).relu()
        t2 = y1 @ w2.double().t() + b2.double()
        y2 = F.group_norm(t2.permute(0, 2, 1), 16, ga2.double(), be2.double(), 1e-5).permute(0, 2, 1).relu()
        o3 = y2 @ w3.double().t() + b3.double()
        refs.append((t1, t2, o3[..., :ccls], o3[..., ccls:] * float(scale)))
        w3s.append(w3.float().contiguous())
        t1d, t2d = torch.full((n, p, 128), float('nan'), device=dev), torch.full((n, p, 128), float('nan'), device=dev)
        pk = [engine_p2.pack_planes_weight(w.reshape(w.shape[0], w.shape[1], 1, 1)).to(dev) for w in (w0, w1, w2, w3)]
        bs = [engine_p2._pad_bias(b, 128).to(dev) for b in (b0, b1, b2, b3)]
        gn = [t.to(dev) for t in (ga1, be1, ga2, be2, scale)]
        xpd = xp.to(dev)
        keep.append((xpd, t1d, t2d, pk, bs, gn))
        a, b_, c = lv0[i], lv1[i], lv2[i]
        a.in_, a.out, a.w0, a.b0, a.w1, a.b1, a.gn_sums = xpd.data_ptr(), t1d.data_ptr(), pk[0].data_ptr(), bs[0].data_ptr(), pk[1].data_ptr(), bs[1].data_ptr(), gsum[0].data_ptr()
        a.in_plane_halfs, a.pixels = xpd[0].numel(), p
        b_.in_, b_.out, b_.w0, b_.b0, b_.gn_sums = t1d.data_ptr(), t2d.data_ptr(), pk[2].data_ptr(), bs[2].data_ptr(), gsum[1].data_ptr()
        b_.gn_in_sums, b_.gn_in_gamma, b_.gn_in_beta, b_.pixels = gsum[0].data_ptr(), gn[0].data_ptr(), gn[1].data_ptr(), p
        c.in_, c.w0, c.b0, c.pixels = t2d.data_ptr(), pk[3].data_ptr(), bs[3].data_ptr(), p
        c.gn_in_sums, c.gn_in_gamma, c.gn_in_beta = gsum[1].data_ptr(), gn[2].data_ptr(), gn[3].data_ptr()
        c.f_out0, c.f_out1, c.scale1 = cls.data_ptr() + poff[i] * ccls * 4, reg.data_ptr() + poff[i] * 16, gn[4].data_ptr()
    # (every level of a launch shares the sums buffer here only because the test gives each level the same image count and
    #  checks levels one at a time: one level per launch)
    for i in range(len(P)):
        gsum.zero_()
        for mode, arr in ((0, lv0), (1, lv1), (2, lv2)):
            d = _lib.PlHeadDesc()
            d.mode, d.n, d.cin, d.relu0, d.gn_in_eps = mode, n, cin, 1, 1e-5
            d.f_c0, d.f_c1, d.f_image_stride0, d.f_image_stride1 = ccls, 4, Ptot * ccls, Ptot * 4
            one = (_lib.PlHeadLevel * 1)(arr[i])
            check(lib().lfd_pl_head_levels(C.byref(d), one, 1, ptr(z), stream_ptr()), 'lfd_pl_head_levels')
        torch.cuda.synchronize()
        t1, t2, rc, rr = refs[i]
        for got, ref, what in ((keep[i][1], t1, 't1'), (keep[i][2], t2, 't2'), (cls[:, poff[i]:poff[i] + P[i]], rc, 'cls'),
                               (reg[:, poff[i]:poff[i] + P[i]], rr, 'reg')):
            gg = got.cpu().double()
            assert not torch.isnan(gg).any(), what + ': unwritten output'
            err, mag = float((gg - ref).abs().max()), float(ref.abs().max())
            print('level %d %s: err %.2e (max |y| %.2f)' % (i, what, err, mag))
            assert err <= 2.5 * TOL * max(1.0, mag), (i, what)
    # all levels in ONE launch per mode (every level its own sums): the level-concatenated outputs again
    gs = torch.zeros((len(P), 2, _lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device=dev)
    for i in range(len(P)):
        lv0[i].gn_sums, lv1[i].gn_in_sums = gs[i, 0].data_ptr(), gs[i, 0].data_ptr()
        lv1[i].gn_sums, lv2[i].gn_in_sums = gs[i, 1].data_ptr(), gs[i, 1].data_ptr()
    c_one, r_one = cls.clone(), reg.clone()
    cls.fill_(float('nan')); reg.fill_(float('nan'))
    for mode, arr in ((0, lv0), (1, lv1), (2, lv2)):
        d = _lib.PlHeadDesc()
        d.mode, d.n, d.cin, d.relu0, d.gn_in_eps = mode, n, cin, 1, 1e-5
        d.f_c0, d.f_c1, d.f_image_stride0, d.f_image_stride1 = ccls, 4, Ptot * ccls, Ptot * 4
        check(lib().lfd_pl_head_levels(C.byref(d), arr, len(P), ptr(z), stream_ptr()), 'lfd_pl_head_levels')
    torch.cuda.synchronize()
    assert torch.equal(cls, c_one) and torch.equal(reg, r_one), 'all levels in one launch != one launch per level'


@pytest.mark.parametrize('sizes,n', [([(19, 37), (3, 5)], 2), ([(8, 8)], 3)])
def test_pl_head_levels_mode3_tower_conv_on_stored_planes_vs_float64(sizes, n):
    """lfd_pl_head_levels mode 3 (heads with separate cls / reg towers, lfd_head.py:88-139 with merge_path_flag False): the first
    tower conv reads the neck's STORED output (planes) -> fp32 out + GroupNorm sums; levels in one launch, against float64."""
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(n + len(sizes))
    z = ops.zero_line(dev)
    arr = (_lib.PlHeadLevel * len(sizes))()
    keep, refs = [], []
    gs = torch.zeros((len(sizes), _lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device=dev)
    for i, (h, w) in enumerate(sizes):
        p = h * w
        x = torch.randn(n, p, 128, generator=g) * 2
        xp = engine_p2.to_planes(x.reshape(n, p, 1, 128))
        xv = engine_p2.from_planes(xp).reshape(n, p, 128).double()
        w1, b1 = torch.randn(128, 128, generator=g) / 128 ** 0.5, torch.randn(128, generator=g)
        refs.append(xv @ w1.double().t() + b1.double())
        out = torch.full((n, p, 128), float('nan'), device=dev)
        pk, bs, xpd = engine_p2.pack_planes_weight(w1.reshape(128, 128, 1, 1)).to(dev), engine_p2._pad_bias(b1, 128).to(dev), xp.to(dev)
        keep.append((out, pk, bs, xpd))
        a = arr[i]
        a.in_, a.out, a.w0, a.b0, a.gn_sums, a.in_plane_halfs, a.pixels = xpd.data_ptr(), out.data_ptr(), pk.data_ptr(), bs.data_ptr(), gs[i].data_ptr(), xpd[0].numel(), p
    d = _lib.PlHeadDesc()
    d.mode, d.n, d.cin = 3, n, 128
    check(lib().lfd_pl_head_levels(C.byref(d), arr, len(sizes), ptr(z), stream_ptr()), 'lfd_pl_head_levels mode 3')
    torch.cuda.synchronize()
    for i, ref in enumerate(refs):
        got = keep[i][0].cpu().double()
        assert not torch.isnan(got).any()
        err, mag = float((got - ref).abs().max()), float(ref.abs().max())
        print('level %d: err %.2e (max |y| %.2f)' % (i, err, mag))
        assert err <= TOL * max(1.0, mag)
        # the fixed-point GroupNorm sums of what was written: per (image, group of 8 channels) sum and sum of squares
        sums = gs[i].sum(0).cpu().double() / 2 ** 24
        r = ref.reshape(n, -1, 16, 8)
        assert torch.allclose(sums[..., 0], r.sum((1, 3)), rtol=1e-6, atol=1e-3) and torch.allclose(sums[..., 1], (r * r).sum((1, 3)), rtol=1e-6, atol=1e-3)
