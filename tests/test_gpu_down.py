"""lfd_downblock_fused_f16 (csrc/down.hip): the first block of a backbone stage (3x3 s2 + 1x1 s2 branch + 3x3 s1 + add) in
one launch.  Gate 1: BIT-IDENTICAL to the two-launch path (lfd_conv2d_downsample_nhwc_f16, then lfd_conv2d_nhwc_f16 with the
branch as residual) for every shape class -- single pixel, strip / segment boundaries, odd sizes, the backbone's maps.
Gate 2: against a float64 convolution of the same fp16 operands (independent of the other kernels)."""
import pytest
import torch
import torch.nn.functional as F

from lfd_amd import ops

pytestmark = pytest.mark.gpu


def _operands(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, h, w, 64, generator=g) * 0.5).half()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().float()
    wd = (torch.randn(64, 64, 1, 1, generator=g) / 8).half().float()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().float()
    b1, bd, b2 = (torch.randn(64, generator=g) * 0.1 for _ in range(3))
    return x, w1, b1, wd, bd, w2, b2


def _both(x, w1, b1, wd, bd, w2, b2):
    xc = x.cuda()
    p1, pd, p2 = (ops.pack_conv_weight(t).cuda() for t in (w1, wd, w2))
    y1, ident = ops.conv2d_downsample_nhwc(xc, p1, b1.cuda(), pd, bd.cuda())
    two = ops.conv2d_nhwc(y1, p2, b2.cuda(), 64, 64, 3, 1, True, residual=ident)
    one = ops.downblock_fused(xc, p1, b1.cuda(), pd, bd.cuda(), p2, b2.cuda())
    torch.cuda.synchronize()
    return one, two


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 2, 3), (3, 5, 7), (1, 8, 16), (1, 9, 17), (2, 16, 32), (1, 7, 15), (2, 37, 45),
                                   (1, 59, 60), (1, 60, 61), (1, 61, 121), (2, 119, 122), (1, 17, 30), (8, 68, 120), (8, 135, 240),
                                   (1, 270, 480), (8, 270, 480), (3, 200, 312), (1, 540, 960), (40, 34, 60)])
def test_fused_downblock_is_bit_identical_to_two_launches(shape):
    one, two = _both(*_operands(*shape, seed=sum(shape)))
    assert one.shape == two.shape
    assert torch.isfinite(one.float()).all()
    bad = (one != two)
    assert not bool(bad.any()), 'mismatches: %d of %d, first at %s' % (int(bad.sum()), bad.numel(), bad.nonzero()[0].tolist())


def test_fused_downblock_is_deterministic_and_repeatable():
    ops_ = _operands(4, 135, 240, 3)
    a, _ = _both(*ops_)
    b, _ = _both(*ops_)
    assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(2, 37, 45), (1, 135, 240)])
def test_fused_downblock_vs_float64(shape):
    x, w1, b1, wd, bd, w2, b2 = _operands(*shape, seed=9)
    one, _ = _both(x, w1, b1, wd, bd, w2, b2)
    x64 = x.float().permute(0, 3, 1, 2).double()
    y1 = F.conv2d(x64, w1.double(), b1.double(), stride=2, padding=1).relu().float().half().double()
    ident = F.conv2d(x64, wd.double(), bd.double(), stride=2).float().half().double()
    ref = (F.conv2d(y1, w2.double(), b2.double(), padding=1) + ident).relu()
    got = one.float().cpu().permute(0, 3, 1, 2).double()
    tol = 1.2e-3 * ref.abs().clamp(min=1.0)
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())


def test_fused_downblock_rejects_aliasing_and_bad_shapes():
    x = torch.zeros(1, 8, 8, 64, dtype=torch.float16).cuda()
    w = torch.zeros(2, 36, 64, 8, dtype=torch.float16).cuda()
    wd = torch.zeros(2, 4, 64, 8, dtype=torch.float16).cuda()
    b = torch.zeros(64).cuda()
    with pytest.raises(RuntimeError):
        ops.downblock_fused(x, w, b, wd, b, w, b, out=x)
    with pytest.raises(RuntimeError):
        ops.downblock_fused(torch.zeros(1, 8, 8, 32, dtype=torch.float16).cuda(), w, b, wd, b, w, b)


@pytest.mark.parametrize('name,shape', [('WIDERFACE_LFD_S', (2, 200, 312)), ('WIDERFACE_LFD_L', (1, 240, 256)), ('TT100K_LFD_L', (1, 192, 320))])
def test_engine_with_and_without_downblock_fusion_agree(monkeypatch, name, shape):
    """whole network: LFD_FUSED_DOWN=1 (every qualifying first block of a stage in one launch) and =0 (two launches) give
    identical logits; the plan marks the blocks it can fuse"""
    from lfd_amd import configs, engine
    outs = []
    n, h, w = shape
    x = (torch.rand(n, h, w, 3, generator=torch.Generator().manual_seed(0)) * 2 - 1).half().cuda()
    for flag in ('1', '0'):
        monkeypatch.setenv('LFD_FUSED_DOWN', flag)
        m = configs.build_model(name)
        configs.perturb_weights(m)
        m.eval().cuda()
        with torch.no_grad():
            outs.append([t.clone() for t in m.forward_resident(x)])
        plan = engine.get_plan(m, m._backbone, m._neck, m._head, x.device)
        marked = [c for c in plan.convs if c.down is not None]
        assert all(c.ds is not None and c.ks == 3 and c.stride == 2 and c.cin == 64 and c.cout == 64 for c in marked)
        if name == 'WIDERFACE_LFD_S':
            assert len(marked) == 3
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
