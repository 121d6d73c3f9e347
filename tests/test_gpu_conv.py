"""Conv / stem kernels (through the C ABI) against a float64 CPU convolution of the same
fp16-rounded operands: differences are fp32 accumulation order + the final fp16 rounding."""
import pytest
import torch
import torch.nn.functional as F

from lfd_amd import engine, ops

pytestmark = pytest.mark.gpu


def _run(cin, cout, ks, s, n, h, w, relu=True, res=False, tail=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half()
    wt = (torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)).half().float()
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=s, padding=ks // 2)
    r = None
    if res:
        r = (torch.randn(n, ref.shape[2], ref.shape[3], cout, generator=g) * 0.5).half()
        ref = ref + r.double().permute(0, 3, 1, 2)
    if relu:
        ref = ref.relu()
    t = None
    if tail:
        w2 = (torch.randn(cout, cout, 1, 1, generator=g) * (1.0 / cout ** 0.5)).half().float()
        b2 = torch.randn(cout, generator=g) * 0.1
        ref = F.conv2d(ref.float().half().double(), w2.double(), b2.double()).relu()
        t = (ops.pack_conv_weight(w2).cuda(), b2.cuda(), True)
    out = ops.conv2d_nhwc(x.cuda(), ops.pack_conv_weight(wt).cuda(), b.cuda(), cin, cout, ks, s, relu,
                          residual=r.cuda() if res else None, tail=t)
    got = out.float().cpu().permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    # fp16 output rounding: half an ulp at the value's magnitude (2^-11 relative) + accumulation noise
    tol = 1.2e-3 * ref.abs().clamp(min=1.0)
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())


@pytest.mark.parametrize('cin,cout,ks,s', [(64, 64, 3, 1), (64, 64, 3, 2), (64, 128, 3, 2), (64, 64, 1, 1), (64, 128, 1, 1),
                                           (64, 64, 1, 2), (64, 128, 1, 2), (128, 128, 3, 1), (128, 128, 3, 2),
                                           (128, 128, 1, 1), (128, 128, 1, 2), (32, 32, 3, 2), (32, 64, 3, 2),
                                           (32, 32, 3, 1), (32, 32, 1, 1), (32, 64, 1, 2)])
def test_conv_variants_odd_sizes(cin, cout, ks, s):
    _run(cin, cout, ks, s, 2, 37, 45)          # odd H/W: tile overhang, zero padding, s2 on odd inputs


@pytest.mark.parametrize('shape', [(1, 1, 1), (3, 5, 7), (1, 17, 30), (2, 135, 240), (1, 68, 120)])
def test_conv3x3_shapes(shape):
    _run(64, 64, 3, 1, *shape)


def test_conv_residual_norelu_tail():
    _run(64, 64, 3, 1, 2, 34, 60, res=True)
    _run(64, 64, 3, 1, 1, 17, 30, relu=False)
    _run(128, 128, 3, 1, 2, 17, 30, res=True)
    _run(64, 64, 3, 2, 2, 41, 51, tail=True)
    _run(32, 32, 3, 2, 2, 41, 51, tail=True)


def test_conv_unsupported_is_loud():
    x = torch.zeros(1, 8, 8, 48, dtype=torch.float16).cuda()
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.conv2d_nhwc(x, torch.zeros(1, 27, 64, 8, dtype=torch.float16).cuda(), torch.zeros(32).cuda(), 48, 32, 3, 1, True)


@pytest.mark.parametrize('c', [32, 64])
@pytest.mark.parametrize('fmt', ['nchw_f32', 'nhwc_f16', 'nhwc_u8'])
@pytest.mark.parametrize('tail', [True, False])
def test_stem_kernel_formats(c, fmt, tail):
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(1)
    n, h, w = 2, 45, 71
    if fmt == 'nhwc_u8':
        img = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
        xin, code = img, 2
        xf = ((img.float() / 255 - 0.5) / 0.5).half().float().permute(0, 3, 1, 2)     # simple_normalize
    else:
        xf = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).half().float()
        xin, code = (xf.contiguous(), 0) if fmt == 'nchw_f32' else (xf.permute(0, 2, 3, 1).contiguous().half(), 1)
    w1 = (torch.randn(c, 3, 3, 3, generator=g) * 0.2).half().float()
    b1 = torch.randn(c, generator=g) * 0.1
    ref = F.conv2d(xf.double(), w1.double(), b1.double(), stride=2, padding=1).relu()
    w2p = b2 = None
    if tail:
        w2 = (torch.randn(c, c, 1, 1, generator=g) * (1 / c ** 0.5)).half().float()
        b2 = (torch.randn(c, generator=g) * 0.1).cuda()
        ref = F.conv2d(ref.float().half().double(), w2.double(), b2.cpu().double()).relu()
        w2p = ops.pack_conv_weight(w2).cuda()
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float16).cuda()
    xin = xin.cuda()
    w1p, b1d = engine.pack_stem_weight(w1).cuda(), b1.cuda()     # keep the device tensors alive across the call
    check(lib().lfd_stem_conv_f16(ptr(xin), code, n, h, w, c, ptr(w1p), ptr(b1d), ptr(w2p), ptr(b2), ptr(out),
                                  stream_ptr()), 'stem')
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2).double()
    tol = 1.2e-3 * ref.abs().clamp(min=1.0)
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())


@pytest.mark.parametrize('cin,cout', [(64, 64), (64, 128), (128, 128), (32, 64)])
def test_conv_with_fused_downsample_branch(cin, cout):
    """3x3 s2 (+ReLU) and the block's 1x1 s2 identity branch (no ReLU) from one launch."""
    import ctypes as C
    from lfd_amd import _lib
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    n, h, w = 2, 37, 45
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half()
    w3 = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (cin * 9) ** 0.5)).half().float()
    w1 = (torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5)).half().float()
    b3, b1 = torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1
    xd = x.float().permute(0, 3, 1, 2).double()
    ref = F.conv2d(xd, w3.double(), b3.double(), stride=2, padding=1).relu()
    refd = F.conv2d(xd, w1.double(), b1.double(), stride=2)
    xg = x.cuda()
    out = torch.empty((n, ref.shape[2], ref.shape[3], cout), dtype=torch.float16).cuda()
    outd = torch.empty_like(out)
    w3p, w1p, b3g, b1g = ops.pack_conv_weight(w3).cuda(), ops.pack_conv_weight(w1).cuda(), b3.cuda(), b1.cuda()
    d = _lib.ConvDesc(n, h, w, cin, cout, 3, 2, 1, 0, 0)
    check(lib().lfd_conv2d_downsample_nhwc_f16(C.byref(d), ptr(xg), ptr(out), ptr(w3p), ptr(b3g), ptr(w1p), ptr(b1g),
                                               ptr(outd), ptr(ops.zero_line(xg.device)), stream_ptr()), 'conv+ds')
    torch.cuda.synchronize()
    for got, exp in ((out, ref), (outd, refd)):
        gd = got.float().cpu().permute(0, 3, 1, 2).double()
        assert bool(((gd - exp).abs() <= 1.2e-3 * exp.abs().clamp(min=1.0)).all()), float((gd - exp).abs().max())


@pytest.mark.parametrize('c', [32, 64])
def test_whole_stem_fused_kernel(c):
    """opt-in single-kernel 'faster' stem (csrc/stem_fused.hip) == the four convs applied in turn."""
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(2)
    n, h, w = 2, 70, 101
    xf = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).half().float()
    ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5),
          (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5)]
    ws = [t.half().float() for t in ws]
    bs = [torch.randn(c, generator=g) * 0.1 for _ in range(4)]
    y = xf.double()
    for wt, b, s, p in zip(ws, bs, (2, 1, 2, 1), (1, 0, 1, 0)):
        y = F.conv2d(y, wt.double(), b.double(), stride=s, padding=p).relu().float().half().double()
    packed = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
    bg = [b.cuda() for b in bs]
    out = torch.empty((n, y.shape[2], y.shape[3], c), dtype=torch.float16).cuda()
    xin = xf.permute(0, 2, 3, 1).contiguous().half().cuda()
    check(lib().lfd_stem_faster_fused_f16(ptr(xin), 1, n, h, w, c, ptr(packed[0]), ptr(bg[0]), ptr(packed[1]), ptr(bg[1]),
                                          ptr(packed[2]), ptr(bg[2]), ptr(packed[3]), ptr(bg[3]), ptr(out), stream_ptr()),
          'fused stem')
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2).double()
    # two fp16-rounded intermediates in the chain: allow a few ulp at the output magnitude
    assert bool(((got - y).abs() <= 4e-3 * y.abs().clamp(min=1.0)).all()), float((got - y).abs().max())


@pytest.mark.parametrize('n,h,w', [(1, 8, 8), (1, 19, 24), (2, 33, 57), (1, 64, 96), (3, 47, 130), (1, 270, 481), (2, 135, 256)])
def test_fused_stem_equals_two_kernel_stem(n, h, w):
    """k_stem2x (whole 'faster' stem in one kernel, NHWC fp16 frames) against the two-kernel stem (k_stem + k_conv with
    a chained 1x1) on the same packed weights: every frame size exercises another combination of tile clipping, raw-row
    misalignment (W * 6 mod 16) and zero padding of the stride-2 intermediate.  Both paths round to fp16 at the same
    three places; they differ only in how the biases enter (fp32 accumulator init vs an fp16 hi+lo MFMA k-step, exact
    to 2^-22) and in summation order, so the outputs agree to about one fp16 ulp."""
    import ctypes as C
    from lfd_amd._lib import ConvDesc, check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(h * 1000 + w)
    c = 64
    ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5),
          (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5)]
    ws = [t.half().float() for t in ws]
    bs = [(torch.randn(c, generator=g) * 0.1).cuda() for _ in range(4)]
    packed = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
    x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda()
    h1, w1 = (h + 1) // 2, (w + 1) // 2
    h2, w2 = (h1 + 1) // 2, (w1 + 1) // 2
    fused = torch.empty((n, h2, w2, c), dtype=torch.float16, device='cuda')
    check(lib().lfd_stem_faster_fused_f16(ptr(x), 1, n, h, w, c, ptr(packed[0]), ptr(bs[0]), ptr(packed[1]), ptr(bs[1]),
                                          ptr(packed[2]), ptr(bs[2]), ptr(packed[3]), ptr(bs[3]), ptr(fused), stream_ptr()), 'fused')
    mid = torch.empty((n, h1, w1, c), dtype=torch.float16, device='cuda')
    check(lib().lfd_stem_conv_f16(ptr(x), 1, n, h, w, c, ptr(packed[0]), ptr(bs[0]), ptr(packed[1]), ptr(bs[1]), ptr(mid), stream_ptr()), 'stem')
    two = torch.empty_like(fused)
    d = ConvDesc(n, h1, w1, c, c, 3, 2, 1, c, 1)
    check(lib().lfd_conv2d_nhwc_f16(C.byref(d), ptr(mid), ptr(two), ptr(packed[2]), ptr(bs[2]), None, ptr(packed[3]), ptr(bs[3]),
                                    ptr(ops.zero_line(x.device)), stream_ptr()), 'conv')
    torch.cuda.synchronize()
    a, b = fused.float(), two.float()
    assert torch.isfinite(a).all()
    tol = 2e-3 * b.abs().clamp(min=1.0)          # 2 fp16 ulp at the value's magnitude
    bad = ((a - b).abs() > tol)
    assert int(bad.sum()) == 0, (int(bad.sum()), float((a - b).abs().max()))
    assert float((a - b).abs().mean()) < 1e-4     # almost all elements identical


@pytest.mark.parametrize('n,h,w', [(1, 8, 8), (1, 19, 24), (1, 64, 96), (2, 135, 256), (3, 200, 312), (1, 1080, 1920)])
def test_fused_stem_aligned_fast_path_equals_general_kernel(n, h, w):
    """k_stem2x<ALN>: frames with 16-byte aligned rows (W % 8 == 0, aligned base) take a kernel whose raw-tile staging
    has the realignment shift and the chunk addresses as constants; the same frame at a 2-byte-misaligned base takes
    the general kernel.  Same arithmetic -> bit-identical outputs (interior and border tiles, tile walk included)."""
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(h * 1000 + w + 7)
    c = 64
    ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5),
          (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5)]
    bs = [(torch.randn(c, generator=g) * 0.1).cuda() for _ in range(4)]
    packed = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
    x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda()
    buf = torch.zeros(x.numel() + 64, dtype=torch.float16, device='cuda')
    xm = buf[1:1 + x.numel()].view_as(x)
    xm.copy_(x)
    assert x.data_ptr() % 16 == 0 and xm.data_ptr() % 16 == 2
    h2, w2 = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
    outs = []
    for t in (x, xm):
        o = torch.full((n, h2, w2, c), float('nan'), dtype=torch.float16, device='cuda')
        check(lib().lfd_stem_faster_fused_f16(ptr(t), 1, n, h, w, c, ptr(packed[0]), ptr(bs[0]), ptr(packed[1]), ptr(bs[1]),
                                              ptr(packed[2]), ptr(bs[2]), ptr(packed[3]), ptr(bs[3]), ptr(o), stream_ptr()), 'fused')
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('n,h,w', [(1, 8, 8), (1, 19, 24), (2, 33, 57), (1, 64, 96), (3, 47, 130), (1, 270, 481), (2, 135, 256),
                                   (1, 1080, 1920), (2, 123, 341)])
def test_fused_stem_uint8_frames_equal_normalised_fp16_frames(n, h, w):
    """k_stem2x<U8>: NHWC uint8 frames with simple_normalize fused into the raw-tile staging == the same kernel on the
    frame normalised on the host ((x/255 - 0.5)/0.5 -> fp16), bit for bit: both build the same LDS image.  Sizes cover
    every byte misalignment of the rows (W * 3 mod 4), clipped tiles and a full 1080p frame (whose byte size is an exact
    multiple of the allocator granule: the clamped edge loads must not run past the buffer)."""
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(h * 1000 + w)
    c = 64
    ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5),
          (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5)]
    bs = [(torch.randn(c, generator=g) * 0.1).cuda() for _ in range(4)]
    packed = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
    img = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    xh = ((img.float() / 255 - 0.5) / 0.5).half().cuda()
    xu = img.cuda()
    h2, w2 = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
    outs = []
    for x, code in ((xh, 1), (xu, 2)):
        o = torch.full((n, h2, w2, c), float('nan'), dtype=torch.float16, device='cuda')
        check(lib().lfd_stem_faster_fused_f16(ptr(x), code, n, h, w, c, ptr(packed[0]), ptr(bs[0]), ptr(packed[1]), ptr(bs[1]),
                                              ptr(packed[2]), ptr(bs[2]), ptr(packed[3]), ptr(bs[3]), ptr(o), stream_ptr()), 'fused')
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


def test_conv128_small_map_split_k_kernel_and_the_generic_kernel():
    """cin = cout = 128, 3x3 stride 1: maps of <= 16384 pixels take the split-K / slab-per-workgroup kernel
    (csrc/conv_small.hip), larger ones the streamed-weight kernel -- both against float64, over partial tiles (17 x 30: one
    valid row in the last tile row, 14 valid columns in the last tile column), with and without residual / ReLU; on the
    same input the two kernels agree to one fp16 ulp (different association of the fp32 sum over k)."""
    import os
    import subprocess
    import sys
    for kw in (dict(res=True), dict(relu=False), dict()):
        _run(128, 128, 3, 1, 1, 17, 30, **kw)
        _run(128, 128, 3, 1, 8, 17, 30, **kw)
    _run(128, 128, 3, 1, 2, 23, 40, res=True)            # TT100K_LFD_L's last stage at 720p
    _run(128, 128, 3, 1, 3, 80, 80, res=True)            # 19200 pixels: the generic kernel
    code = '''
import torch, sys
sys.path.insert(0, %r)
from lfd_amd import ops
g = torch.Generator().manual_seed(4)
x = (torch.randn(8, 17, 30, 128, generator=g) * 0.5).half().cuda()
w = ops.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) / 34).half().float()).cuda()
b = (torch.randn(128, generator=g) * 0.1).cuda()
y = ops.conv2d_nhwc(x, w, b, 128, 128, 3, 1, True, residual=x)
torch.save(y.cpu(), sys.argv[1])
''' % os.path.dirname(os.path.dirname(ops.__file__))
    import tempfile
    outs = []
    for flag in ('1', '0'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            r = subprocess.run([sys.executable, '-c', code, f.name], env=dict(os.environ, LFD_CONV128_SPLITK=flag), capture_output=True,
                               text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(f.name).float())
    d = (outs[0] - outs[1]).abs()
    ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(outs[0].abs(), outs[1].abs()).clamp(min=2.0 ** -14))) - 10)
    # (below 2^-14 the fp16 spacing, 2^-24, is finer than the fp32 accumulation noise of a sum of O(1) terms: + 2^-22)
    assert bool((d <= ulp + 2.0 ** -22).all()) and float((d > 0).float().mean()) < 0.01, (float(d.max()), float((d > 0).float().mean()))
