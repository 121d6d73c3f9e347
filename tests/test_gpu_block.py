"""lfd_fasterblock_fused_f16 (csrc/block.hip): a whole FasterBlock without downsample branch in one launch.
Gate 1: BIT-IDENTICAL to the two-launch path (lfd_conv2d_nhwc_f16 twice: same rounding points, same k order) for every
shape class -- single pixel, tile-boundary sizes, odd sizes, the backbone's 135x240 / 68x120 / 34x60 maps at batch 8.
Gate 2: against a float64 convolution of the same fp16 operands (independent of the other kernel)."""
import pytest
import torch
import torch.nn.functional as F

from lfd_amd import ops

pytestmark = pytest.mark.gpu


def _operands(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, h, w, 64, generator=g) * 0.5).half()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().float()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().float()
    b1 = torch.randn(64, generator=g) * 0.1
    b2 = torch.randn(64, generator=g) * 0.1
    return x, w1, b1, w2, b2


def _both(x, w1, b1, w2, b2):
    xc = x.cuda()
    p1, p2 = ops.pack_conv_weight(w1).cuda(), ops.pack_conv_weight(w2).cuda()
    mid = ops.conv2d_nhwc(xc, p1, b1.cuda(), 64, 64, 3, 1, True)
    two = ops.conv2d_nhwc(mid, p2, b2.cuda(), 64, 64, 3, 1, True, residual=xc)
    one = ops.fasterblock_fused(xc, p1, b1.cuda(), p2, b2.cuda())
    torch.cuda.synchronize()
    return one, two


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 2, 3), (3, 5, 7), (1, 8, 16), (1, 9, 17), (2, 16, 32), (1, 7, 15), (2, 37, 45),
                                   (1, 17, 30), (8, 34, 60), (8, 68, 120), (8, 135, 240), (2, 180, 320), (1, 270, 480)])
def test_fused_block_is_bit_identical_to_two_launches(shape):
    one, two = _both(*_operands(*shape, seed=sum(shape)))
    assert torch.isfinite(one.float()).all()
    bad = (one != two)
    assert not bool(bad.any()), 'mismatches: %d of %d, first at %s' % (int(bad.sum()), bad.numel(), bad.nonzero()[0].tolist())


@pytest.mark.parametrize('mode', ['0', '1'])
def test_both_block_kernels_are_bit_identical_to_two_launches(mode):
    """lfd_fasterblock_fused_f16 picks the 8 x 16-tile kernel (csrc/block.hip) or the row-streaming kernel (csrc/block_rows.hip)
    by map size; LFD_BLOCK_ROWS forces one of them -- read once per process, so each mode runs in a fresh interpreter"""
    import os, subprocess, sys
    code = '''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_block as t
for shape in [(1, 1, 1), (1, 2, 3), (3, 5, 7), (1, 8, 16), (1, 9, 17), (2, 16, 32), (1, 7, 15), (2, 37, 45), (1, 31, 61), (2, 29, 90),
              (1, 17, 30), (8, 34, 60), (8, 68, 120), (8, 135, 240), (1, 270, 480), (40, 20, 33)]:
    one, two = t._both(*t._operands(*shape, seed=sum(shape)))
    assert torch.isfinite(one.float()).all() and torch.equal(one, two), shape
print('ok')
''' % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lfd-a-light-and-fast-detector_amd'))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, LFD_BLOCK_ROWS=mode), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_block_is_deterministic_and_repeatable():
    ops_ = _operands(4, 68, 120, 3)
    a, _ = _both(*ops_)
    b, _ = _both(*ops_)
    assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(2, 37, 45), (1, 68, 120)])
def test_fused_block_vs_float64(shape):
    x, w1, b1, w2, b2 = _operands(*shape, seed=9)
    one, _ = _both(x, w1, b1, w2, b2)
    x64 = x.float().permute(0, 3, 1, 2).double()
    mid = F.conv2d(x64, w1.double(), b1.double(), padding=1).relu().float().half().double()
    ref = (F.conv2d(mid, w2.double(), b2.double(), padding=1) + x64).relu()
    got = one.float().cpu().permute(0, 3, 1, 2).double()
    tol = 1.2e-3 * ref.abs().clamp(min=1.0)
    assert bool(((got - ref).abs() <= tol).all()), float((got - ref).abs().max())


def test_fused_block_rejects_aliasing_and_bad_shapes():
    x = torch.zeros(1, 8, 8, 64, dtype=torch.float16).cuda()
    w = torch.zeros(2, 36, 64, 8, dtype=torch.float16).cuda()
    b = torch.zeros(64).cuda()
    with pytest.raises(RuntimeError):
        ops.fasterblock_fused(x, w, b, w, b, out=x)
    with pytest.raises(RuntimeError):
        ops.fasterblock_fused(torch.zeros(1, 8, 8, 32, dtype=torch.float16).cuda(), w, b, w, b)


def test_engine_with_and_without_block_fusion_agree(monkeypatch):
    """whole network: LFD_FUSED_BLOCK=0 (two launches per block) and the default give identical logits"""
    from lfd_amd import configs
    outs = []
    x = (torch.rand(2, 200, 312, 3, generator=torch.Generator().manual_seed(0)) * 2 - 1).half().cuda()
    for flag in ('1', '0'):
        monkeypatch.setenv('LFD_FUSED_BLOCK', flag)
        m = configs.build_model('WIDERFACE_LFD_S')
        configs.perturb_weights(m)
        m.eval().cuda()
        with torch.no_grad():
            outs.append([t.clone() for t in m.forward_resident(x)])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ---------------------------------------------------------------- 128 channels, small maps (csrc/block128.hip)
def _operands128(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, h, w, 128, generator=g) * 0.5).half()
    w1 = (torch.randn(128, 128, 3, 3, generator=g) / 34).half().float()
    w2 = (torch.randn(128, 128, 3, 3, generator=g) / 34).half().float()
    b1 = torch.randn(128, generator=g) * 0.1
    b2 = torch.randn(128, generator=g) * 0.1
    return x, w1, b1, w2, b2


def _both128(x, w1, b1, w2, b2):
    xc = x.cuda()
    p1, p2 = ops.pack_conv_weight(w1).cuda(), ops.pack_conv_weight(w2).cuda()
    mid = ops.conv2d_nhwc(xc, p1, b1.cuda(), 128, 128, 3, 1, True)
    two = ops.conv2d_nhwc(mid, p2, b2.cuda(), 128, 128, 3, 1, True, residual=xc)
    one = ops.fasterblock128_fused(xc, p1, b1.cuda(), p2, b2.cuda())
    torch.cuda.synchronize()
    return one, two, mid


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 2, 3), (3, 5, 7), (1, 4, 8), (1, 5, 9), (2, 8, 16), (1, 7, 15), (1, 17, 30), (8, 17, 30),
                                   (8, 12, 20), (1, 34, 60), (16, 23, 40)])
def test_fused_block128_is_bit_identical_to_two_launches(shape):
    """the two launches run the split-K kernel (n * h * w <= 16384): same quarters of K, added in the same order"""
    one, two, _ = _both128(*_operands128(*shape, seed=sum(shape)))
    assert torch.isfinite(one.float()).all()
    bad = (one != two)
    assert not bool(bad.any()), 'mismatches: %d of %d, first at %s' % (int(bad.sum()), bad.numel(), bad.nonzero()[0].tolist())


def test_fused_block128_against_float64():
    x, w1, b1, w2, b2 = _operands128(2, 17, 30, seed=5)
    one, _, _ = _both128(x, w1, b1, w2, b2)
    xd = x.double().permute(0, 3, 1, 2)
    mid = F.relu(F.conv2d(xd, w1.double(), b1.double(), padding=1)).half().double()
    ref = F.relu(F.conv2d(mid, w2.double(), b2.double(), padding=1) + xd).permute(0, 2, 3, 1)
    err = (one.cpu().double() - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err


def test_fused_block128_rejects_bad_arguments():
    w = torch.zeros(4 * 72 * 64 * 8, dtype=torch.float16).cuda()
    b = torch.zeros(128).cuda()
    x = torch.zeros(1, 8, 8, 128, dtype=torch.float16).cuda()
    with pytest.raises(RuntimeError):
        ops.fasterblock128_fused(x, w, b, w, b, out=x)
    with pytest.raises(RuntimeError):
        ops.fasterblock128_fused(torch.zeros(1, 8, 8, 64, dtype=torch.float16).cuda(), w, b, w, b)


@pytest.mark.parametrize('shape', [(2, 200, 312), (8, 1080, 1920)])
def test_engine_with_and_without_block128_fusion_agree(monkeypatch, shape):
    """whole network: LFD_FUSED_BLOCK128=0 (two split-K launches per 128-channel block) and the default (one launch) give
    identical logits, at a small shape and at the headline batch"""
    from lfd_amd import configs, engine
    outs = []
    x = (torch.rand(shape[0], shape[1], shape[2], 3, generator=torch.Generator().manual_seed(0)) * 2 - 1).half().cuda()
    for flag in ('1', '0'):
        monkeypatch.setenv('LFD_FUSED_BLOCK128', flag)
        assert engine._use_fused_block128(8, 17, 30) == (flag == '1')
        m = configs.build_model('WIDERFACE_LFD_S')
        configs.perturb_weights(m)
        m.eval().cuda()
        with torch.no_grad():
            outs.append([t.clone() for t in m.forward_resident(x)])
        del m
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
