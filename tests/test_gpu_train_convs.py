"""Training-mode conv stack kernels (csrc/train.hip) against plain PyTorch fp32 references of the
same ops on the GPU: batch-norm statistics / apply / backward, stride-2 zero insertion, MFMA weight gradient, the
3-channel first conv, the data gradient through the forward conv kernel, and the whole backbone forward + backward
against the torch modules (nn.Conv2d / nn.BatchNorm2d autograd = what the reference trains through).
Tolerances: activations are stored in fp16 (2^-11 relative), accumulation is fp32."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lfd_amd import configs, ops, train_engine

pytestmark = pytest.mark.gpu


def _rand16(shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(shape, generator=g, device='cuda') * scale + shift).half()


def _nchw(t):
    return t.permute(0, 3, 1, 2).float()


def _rel(a, b):
    return float((a.detach() - b.detach()).norm() / (b.detach().norm() + 1e-30))


@pytest.mark.parametrize('c', [32, 64, 128])
def test_bn_train_stats_and_apply(c):
    y = _rand16((3, 33, 47, c), c, 2.0, 0.7)
    bn = torch.nn.BatchNorm2d(c).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.1)
        bn.running_var.uniform_(0.5, 2.0)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    res = _rand16((3, 33, 47, c), 1)
    ref = F.relu(bn(_nchw(y)) + _nchw(res))                 # updates bn.running_*
    stats = ops.bn_train_stats(y, bn.eps, bn.momentum, rm, rv)
    yd = y.double().reshape(-1, c)
    torch.testing.assert_close(stats[:c].double(), yd.mean(0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(stats[c:].double(), 1 / torch.sqrt(yd.var(0, unbiased=False) + bn.eps), rtol=1e-5, atol=0)
    torch.testing.assert_close(rm, bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, bn.running_var, rtol=1e-5, atol=1e-6)
    z = ops.bn_train_apply(y, stats, bn.weight.detach(), bn.bias.detach(), res, True)
    torch.testing.assert_close(_nchw(z), ref, rtol=2e-3, atol=2e-3)
    z2 = ops.bn_train_apply(y, stats, bn.weight.detach(), bn.bias.detach(), None, False)
    torch.testing.assert_close(_nchw(z2), bn(_nchw(y)), rtol=2e-3, atol=2e-3)


# (cin, cout, ks, stride): the seven shapes with statistics in the conv epilogue, then two that run conv + statistics pass
_STATS_SHAPES = [(64, 64, 3, 1), (64, 64, 3, 2), (64, 128, 3, 2), (64, 64, 1, 1), (64, 128, 1, 1), (64, 64, 1, 2), (64, 128, 1, 2),
                 (128, 128, 1, 1), (128, 128, 3, 1), (32, 64, 3, 2)]


@pytest.mark.parametrize('cin,cout,ks,stride', _STATS_SHAPES)
@pytest.mark.parametrize('nhw', [(2, 37, 53), (1, 8, 8), (3, 130, 70)])
def test_conv_with_bn_statistics_in_the_epilogue(cin, cout, ks, stride, nhw):
    """lfd_conv2d_bn_stats_nhwc_f16 (csrc/conv_stats.hip): y bit-identical to lfd_conv2d_nhwc_f16; mean / rstd / running
    statistics equal to fp64 sums over the STORED fp16 y (what F.batch_norm(training=True) normalises with), i.e. to
    lfd_bn_train_stats_f16 up to the order of the fp32 partial sums.  Ragged maps: tiles hang over the right / bottom edge."""
    n, h, w = nhw
    x = _rand16((n, h, w, cin), 7 * cin + ks + stride, 1.0, 0.2)
    g = torch.Generator(device='cuda').manual_seed(cout + ks)
    wt = torch.randn((cout, cin, ks, ks), generator=g, device='cuda') * (2.0 / (cin * ks * ks)) ** 0.5
    pk = ops.pack_conv_weight_train(wt)
    zb = torch.zeros(cout, device='cuda')
    y0 = ops.conv2d_nhwc(x, pk, zb, cin, cout, ks, stride, False)
    rm = torch.randn(cout, generator=g, device='cuda') * 0.1
    rv = torch.rand(cout, generator=g, device='cuda') + 0.5
    rm0, rv0, rm1, rv1 = rm.clone(), rv.clone(), rm.clone(), rv.clone()
    st0 = ops.bn_train_stats(y0, 1e-5, 0.1, rm0, rv0)
    y1, st1 = ops.conv2d_bn_stats(x, pk, zb, cin, cout, ks, stride, 1e-5, 0.1, rm1, rv1)
    assert torch.equal(y0, y1)
    yd = y0.double().reshape(-1, cout)
    torch.testing.assert_close(st1[:cout].double(), yd.mean(0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st1[cout:].double(), 1 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5), rtol=1e-5, atol=0)
    torch.testing.assert_close(st1, st0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rm1, rm0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv1, rv0, rtol=1e-5, atol=1e-6)
    y2, st2 = ops.conv2d_bn_stats(x, pk, zb, cin, cout, ks, stride, 1e-5, 0.1)      # no running statistics; same bits again
    assert torch.equal(y2, y1) and torch.equal(st2, st1)


def _bn_pass_outputs(seed=3):
    """apply + backward of a BatchNorm unit on seeded tensors (ragged sizes, with and without residual / ReLU): every output"""
    outs = []
    for c, shape, relu, with_res in [(64, (3, 33, 47), True, True), (128, (2, 17, 30), True, False), (32, (1, 5, 3), False, False),
                                     (64, (4, 160, 160), True, False)]:
        y = _rand16(shape + (c,), seed + c, 2.0, 0.7)
        dz = _rand16(shape + (c,), seed + 1, 0.05)
        res = _rand16(shape + (c,), seed + 2) if with_res else None
        g = torch.Generator(device='cuda').manual_seed(seed)
        gamma = torch.rand(c, generator=g, device='cuda') + 0.5
        beta = torch.randn(c, generator=g, device='cuda') * 0.3
        stats = ops.bn_train_stats(y, 1e-5, 0.1)
        z = ops.bn_train_apply(y, stats, gamma, beta, res, relu)
        dgamma, dbeta = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        dy, gg = ops.bn_train_backward(dz, y, z if with_res else None, stats, gamma, 1.0 / 1024, dgamma, dbeta,
                                       want_g=with_res, accumulate=True, relu=relu, beta=beta)
        outs += [z, dy, dgamma, dbeta] + ([gg] if gg is not None else [])
    return outs


@pytest.mark.parametrize('c,relu,with_res', [(64, True, True), (128, True, False), (32, False, False)])
def test_bn_train_backward_vs_autograd(c, relu, with_res):
    shape = (2, 29, 41, c)
    y = _rand16(shape, 3, 1.5, 0.2)
    res = _rand16(shape, 4) if with_res else None
    dz = _rand16(shape, 5, 0.02)
    gamma = torch.empty(c, device='cuda').uniform_(0.5, 1.5)
    beta = torch.empty(c, device='cuda').normal_(0, 0.3)
    yr = _nchw(y).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = _nchw(res).requires_grad_(True) if with_res else None
    o = F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)
    if with_res:
        o = o + rr
    if relu:
        o = F.relu(o)
    o.backward(_nchw(dz))
    stats = ops.bn_train_stats(y, 1e-5, 0.1)
    z = ops.bn_train_apply(y, stats, gamma, beta, res, relu)
    dgamma, dbeta = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    scale = 8.0                                  # pretend dz carries a loss scale of 8: parameter gradients come back unscaled
    dy, g = ops.bn_train_backward((dz.float() * scale).half(), y, z if relu else None, stats, gamma, 1 / scale, dgamma,
                                  dbeta, want_g=with_res)
    ref_dy = yr.grad * scale
    err = (_nchw(dy) - ref_dy).abs().max() / ref_dy.abs().max()
    assert float(err) < 3e-3, float(err)
    torch.testing.assert_close(dgamma, gr.grad, rtol=3e-3, atol=3e-3 * float(gr.grad.abs().max()))
    torch.testing.assert_close(dbeta, br.grad, rtol=3e-3, atol=3e-3 * float(br.grad.abs().max()))
    if with_res:
        torch.testing.assert_close(_nchw(g), rr.grad * scale, rtol=2e-3, atol=1e-4)
    if relu and not with_res:      # the same backward with the ReLU mask recomputed from y instead of read from z
        dg2, db2 = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
        dy2, _ = ops.bn_train_backward((dz.float() * scale).half(), y, None, stats, gamma, 1 / scale, dg2, db2, relu=True,
                                       beta=beta)
        assert float((_nchw(dy2) - ref_dy).abs().max() / ref_dy.abs().max()) < 3e-3
        torch.testing.assert_close(dg2, gr.grad, rtol=3e-3, atol=3e-3 * float(gr.grad.abs().max()))
        torch.testing.assert_close(db2, br.grad, rtol=3e-3, atol=3e-3 * float(br.grad.abs().max()))


@pytest.mark.parametrize('relu', [True, False])
def test_gn_train_forward_backward_vs_autograd(relu):
    n, h, w, c, groups = 3, 37, 29, 128, 16
    y = _rand16((n, h, w, c), 8, 1.5, 0.3)
    dz = _rand16((n, h, w, c), 9, 0.02)
    gamma = torch.empty(c, device='cuda').uniform_(0.5, 1.5)
    beta = torch.empty(c, device='cuda').normal_(0, 0.3)
    yr = _nchw(y).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    o = F.group_norm(yr, groups, gr, br, 1e-5)
    if relu:
        o = F.relu(o)
    o.backward(_nchw(dz))
    stats = ops.gn_train_stats(y, groups, 1e-5)
    yd = _nchw(y).double().reshape(n, groups, -1)
    torch.testing.assert_close(stats.view(n, 2, groups)[:, 0].double(), yd.mean(2), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(stats.view(n, 2, groups)[:, 1].double(), 1 / torch.sqrt(yd.var(2, unbiased=False) + 1e-5),
                               rtol=1e-5, atol=0)
    z = ops.gn_train_apply(y, groups, stats, gamma, beta, relu)
    torch.testing.assert_close(_nchw(z), o.detach(), rtol=2e-3, atol=2e-3)
    # the two-launch form (round 4: the apply pass adds the partial rows itself): the same statistics and output bit for bit
    stats2, z2 = ops.gn_train_stats_apply(y, groups, 1e-5, gamma, beta, relu)
    assert torch.equal(stats2, stats) and torch.equal(z2, z)
    dgamma, dbeta = torch.full((c,), 2.0, device='cuda'), torch.full((c,), -1.0, device='cuda')
    scale = 16.0
    dy = ops.gn_train_backward((dz.float() * scale).half(), y, z if relu else None, groups, stats, gamma, 1 / scale, dgamma,
                               dbeta, accumulate=True)
    assert _rel(_nchw(dy) / scale, yr.grad) < 3e-3
    assert _rel(dgamma - 2.0, gr.grad) < 3e-3 and _rel(dbeta + 1.0, br.grad) < 3e-3      # += into the existing buffers


@pytest.mark.parametrize('cout,cin,ks', [(64, 64, 3), (128, 64, 3), (64, 128, 1), (32, 32, 3), (128, 128, 1)])
def test_pack_kernel_equals_host_pack(cout, cin, ks):
    w = torch.randn(cout, cin, ks, ks, device='cuda')
    assert torch.equal(ops.pack_conv_weight_train(w), ops.pack_conv_weight(w))
    assert torch.equal(ops.pack_conv_weight_train(w, data_gradient=True),
                       ops.pack_conv_weight(w.permute(1, 0, 2, 3).flip(2, 3).contiguous()))
    if ks == 1:
        wp = torch.zeros(64 if cout <= 64 else 128, cin, 1, 1, device='cuda')
        wp[:5] = w[:5]
        assert torch.equal(ops.pack_conv_weight_train(w[:5].contiguous(), rows=wp.size(0)), ops.pack_conv_weight(wp))


def test_zero_insert2_exact():
    for hi, wi, ho, wo in [(5, 7, 10, 14), (5, 7, 9, 13), (1, 1, 1, 1), (34, 60, 68, 120)]:
        t = _rand16((2, hi, wi, 64), hi)
        o = ops.zero_insert2(t, ho, wo)
        ref = torch.zeros((2, ho, wo, 64), dtype=torch.float16, device='cuda')
        ref[:, ::2, ::2] = t[:, :(ho + 1) // 2, :(wo + 1) // 2]
        assert torch.equal(o, ref)


@pytest.mark.parametrize('ks,stride', [(3, 1), (3, 2), (1, 1), (1, 2)])
@pytest.mark.parametrize('cin,cout', [(64, 64), (64, 128), (128, 128), (32, 32), (32, 64), (128, 64)])
def test_conv_wgrad_vs_autograd(ks, stride, cin, cout):
    n, h, w = 2, 37, 53
    x = _rand16((n, h, w, cin), 10 + cin)
    ho, wo = (h + 2 * (ks // 2) - ks) // stride + 1, (w + 2 * (ks // 2) - ks) // stride + 1
    dy = _rand16((n, ho, wo, cout), 20 + cout, 0.05)
    wt = torch.zeros(cout, cin, ks, ks, device='cuda', requires_grad=True)
    F.conv2d(_nchw(x), wt, None, stride, ks // 2).backward(_nchw(dy))
    dw = ops.conv_wgrad(x, dy, ks, stride, 0.25)
    assert dw.shape == wt.shape
    err = (dw * 4 - wt.grad).abs().max() / wt.grad.abs().max()
    assert float(err) < 2e-3, float(err)


def test_conv_wgrad_large_map_and_determinism():
    x = _rand16((4, 80, 80, 64), 1)
    dy = _rand16((4, 80, 80, 64), 2, 0.05)
    wt = torch.zeros(64, 64, 3, 3, device='cuda', requires_grad=True)
    F.conv2d(_nchw(x), wt, None, 1, 1).backward(_nchw(dy))
    a, b = ops.conv_wgrad(x, dy, 3, 1, 1.0), ops.conv_wgrad(x, dy, 3, 1, 1.0)
    assert torch.equal(a, b)
    assert float((a - wt.grad).abs().max() / wt.grad.abs().max()) < 2e-3


@pytest.mark.parametrize('c', [32, 64])
def test_first_stem_conv_forward_and_wgrad(c):
    g = torch.Generator(device='cuda').manual_seed(c)
    x = torch.randn((3, 3, 61, 77), generator=g, device='cuda')
    wt = (torch.randn((c, 3, 3, 3), generator=g, device='cuda') * 0.2).requires_grad_(True)
    ref = F.conv2d(x, wt, None, 2, 1)
    y = ops.stem_conv0_train_fwd(x, wt)
    assert y.shape == (3, 31, 39, c)
    torch.testing.assert_close(_nchw(y), ref, rtol=3e-3, atol=4e-3)      # image and weights rounded to fp16 for the MFMA
    dy = _rand16((3, 31, 39, c), 7, 0.05)
    ref.backward(_nchw(dy))
    dw = ops.stem_conv0_wgrad(x, dy, 0.5)
    assert float((dw * 2 - wt.grad).abs().max() / wt.grad.abs().max()) < 2e-3


@pytest.mark.parametrize('c', [32, 64])
@pytest.mark.parametrize('nhw', [(3, 61, 77), (2, 256, 320), (1, 5, 3)])
def test_first_stem_conv_with_bn_statistics(c, nhw):
    """lfd_stem_conv0_train_fwd_bn_stats: y bit-identical to lfd_stem_conv0_train_fwd, statistics = fp64 sums over the stored
    fp16 y (64 channels: from the conv's own stores; 32: the separate pass)"""
    n, h, w = nhw
    g = torch.Generator(device='cuda').manual_seed(c + h)
    x = torch.randn((n, 3, h, w), generator=g, device='cuda')
    wt = torch.randn((c, 3, 3, 3), generator=g, device='cuda') * 0.2
    y0 = ops.stem_conv0_train_fwd(x, wt)
    rm = torch.randn(c, generator=g, device='cuda') * 0.1
    rv = torch.rand(c, generator=g, device='cuda') + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    st0 = ops.bn_train_stats(y0, 1e-5, 0.1, rm0, rv0)
    y1, st1 = ops.stem_conv0_train_fwd_bn_stats(x, wt, 1e-5, 0.1, rm, rv)
    assert torch.equal(y0, y1)
    yd = y0.double().reshape(-1, c)
    torch.testing.assert_close(st1[:c].double(), yd.mean(0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st1[c:].double(), 1 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5), rtol=1e-5, atol=0)
    torch.testing.assert_close(st1, st0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rm, rm0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, rv0, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('ks,stride,cin,cout', [(3, 1, 64, 64), (3, 2, 64, 128), (1, 2, 64, 64), (3, 2, 128, 128),
                                                (1, 1, 32, 32), (3, 2, 32, 64), (1, 2, 32, 64), (1, 2, 64, 128),
                                                (1, 2, 128, 128), (3, 1, 128, 128), (3, 2, 32, 32), (1, 1, 64, 64)])
def test_data_gradient_through_the_forward_conv_kernel(ks, stride, cin, cout):
    """dx = conv(zero_insert(dy), W^T with flipped taps) (+ already collected gradient) == autograd's input gradient"""
    n, h, w = 2, 37, 53
    g = torch.Generator(device='cuda').manual_seed(ks * 10 + stride)
    wt = (torch.randn((cout, cin, ks, ks), generator=g, device='cuda') * 0.05).half().float()
    x = _nchw(_rand16((n, h, w, cin), 1)).requires_grad_(True)
    yref = F.conv2d(x, wt, None, stride, ks // 2)
    dy = _rand16((n, yref.size(2), yref.size(3), cout), 2, 0.05)
    yref.backward(_nchw(dy))
    acc = _rand16((n, h, w, cin), 3, 0.05)
    d = ops.zero_insert2(dy, h, w) if stride == 2 else dy
    zb = torch.zeros(cin, device='cuda')
    dx = ops.conv2d_nhwc(d, train_engine._dgrad_weight(wt), zb, cout, cin, ks, 1, False, residual=acc)
    ref = x.grad + _nchw(acc)
    assert float((_nchw(dx) - ref).abs().max() / ref.abs().max()) < 3e-3


def _cos(a, b):
    a, b = a.detach(), b.detach()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def _mask_replay_reference(units, acts, x, outs=None):
    """The whole forward as PyTorch fp32 autograd over the SAME unit graph (conv -> BatchNorm / GroupNorm with batch statistics
    -> (+ residual) -> ReLU), with every ReLU taking its MASK from the tensor the HIP path stored for that unit.  A
    reduced-precision forward flips ~0.5 % of the ReLU decisions of a 30-layer network against an fp32 forward, and a flipped
    mask is a 100 % error of that element's gradient: replaying the masks removes exactly that effect and leaves what
    the backward kernels are responsible for.  -> ({activation index: NCHW tensor}, {id(param): leaf})"""
    leaves = {}

    def leaf(p):
        if id(p) not in leaves:
            leaves[id(p)] = p.detach().clone().float().requires_grad_(True)
        return leaves[id(p)]
    ref = {}
    for u in units:
        xin = x if u.first else ref[u.src]
        y = F.conv2d(xin, leaf(u.conv.weight), None, u.conv.stride, u.conv.padding)
        if isinstance(u.norm, torch.nn.GroupNorm):
            z = F.group_norm(y, u.norm.num_groups, leaf(u.norm.weight), leaf(u.norm.bias), u.norm.eps)
        else:
            z = F.batch_norm(y, None, None, leaf(u.norm.weight), leaf(u.norm.bias), True, 0.1, u.norm.eps)
        if u.res is not None:
            z = z + ref[u.res]
        if u.relu:
            z = z * (_nchw(acts[u.dst]) > 0).float()
        ref[u.dst] = z
    return ref, leaves


def _check_units_against_autograd(units, acts, tape, trace, S):
    """every recorded unit against PyTorch autograd of that unit GIVEN the tensors the HIP path stored"""
    for rec in trace:
        u = units[rec['ui']]
        y = _nchw(tape[rec['ui']][0]).requires_grad_(True)
        gam, bet = u.norm.weight.detach().clone().requires_grad_(True), u.norm.bias.detach().clone().requires_grad_(True)
        res = _nchw(acts[u.res]).requires_grad_(True) if u.res is not None else None
        if isinstance(u.norm, torch.nn.GroupNorm):
            z = F.group_norm(y, u.norm.num_groups, gam, bet, u.norm.eps)
        else:
            z = F.batch_norm(y, None, None, gam, bet, True, 0.1, u.norm.eps)
        if res is not None:
            z = z + res
        k = 'unit %d' % rec['ui']
        keep = torch.ones_like(z)
        if u.relu:
            # Knife-edge ReLU inputs: |z| within 1e-3 of the tensor's rms, where the sign is decided by the order of the fp32
            # operations (PyTorch's batch_norm here, y * a + b in the forward kernel, gamma * xhat + beta in the backward one).
            # One such element of 98,304 deciding the other way is a 100 % error of its own gradient and was the WHOLE 1.5e-3
            # of a unit whose other elements agree to 2.1e-4 (tools/timing/unit_dy_probe.py).  They must be rare, every other
            # element must decide like the stored output, and the gradient comparison leaves the knife-edge elements out --
            # which makes room for a gate at fp16-storage accuracy instead of one with a flipped sign priced in.
            zd = z.detach()
            knife = zd.abs() < 1e-3 * zd.pow(2).mean().sqrt()
            assert float(knife.float().mean()) < 5e-3, k
            assert torch.equal((zd > 0) | knife, (_nchw(acts[u.dst]) > 0) | knife), k
            keep = (~knife).float()
            z = F.relu(z)
        z.backward(_nchw(rec['dz']) / S)
        gate = 3e-3 if isinstance(u.norm, torch.nn.GroupNorm) else 1e-3      # (BatchNorm units measure 2.0-2.2e-4)
        assert _rel(_nchw(rec['dy']) / S * keep, y.grad * keep) < gate, k
        if not isinstance(u.norm, torch.nn.GroupNorm):          # GroupNorm buffers accumulate over the levels: checked in (c)
            # sums of random-sign fp16-rounded terms: the rounding noise does not cancel like the signal does
            assert _rel(rec['dgamma'], gam.grad) < 1.5e-2 and _rel(rec['dbeta'], bet.grad) < 1.5e-2, k
        if res is not None:
            assert _rel(_nchw(rec['g']) / S * keep, res.grad * keep) < 1e-3, k
        xin = acts[u.src] if u.first else _nchw(acts[u.src])
        xin = xin.detach().clone().requires_grad_(not u.first)
        wt = u.conv.weight.detach().clone().requires_grad_(True)
        F.conv2d(xin, wt, None, u.conv.stride, u.conv.padding).backward(_nchw(rec['dy']) / S)
        assert _rel(rec['dw'], wt.grad) < 3e-3, k
        if not u.first:
            ref_dx = xin.grad + (_nchw(rec['dx_prev']) / S if rec['dx_prev'] is not None else 0)
            assert _rel(_nchw(rec['dx']) / S, ref_dx) < 3e-3, k


@pytest.mark.parametrize('name,hw', [('WIDERFACE_LFD_XS', (160, 192)), ('WIDERFACE_LFD_S', (128, 160)),
                                     ('TT100K_LFD_L', (96, 128)), ('WIDERFACE_LFD_L', (64, 96))])
def test_backbone_train_forward_backward_vs_torch_modules(name, hw):
    """One training forward + backward of the whole backbone on the HIP kernels vs PyTorch.
    (a) forward vs the same nn.Modules in fp32: features and running statistics to fp16-storage accuracy;
    (b) backward, every unit against PyTorch autograd of that unit GIVEN the tensors the HIP path stored (its input,
        pre-norm output, incoming gradient): dy, dgamma, dbeta, dW, dx incl. the accumulation over consumers -- tight;
    (c) end to end vs fp32 autograd of the modules: loose by nature -- a 1e-2 forward difference after 30 fp16-stored
        layers flips the ReLU mask of ~0.5 % of the activations, which alone is a sqrt(0.005) = 7 % relative L2 effect per
        layer on a random upstream gradient (measured: layer 29 receives dz exact to 2e-4 and returns dy off by 6.8 %
        purely through its mask).  Any fp16 / bf16 training path has this property; it is not a kernel error, which
        (b) shows."""
    torch.manual_seed(1)
    ma = configs.build_model(name).cuda().train()
    configs.perturb_weights(ma)
    mc = copy.deepcopy(ma)
    x = torch.randn(4, 3, hw[0], hw[1], device='cuda')
    assert train_engine.supported(ma._backbone)
    units, taps = train_engine.build_units(ma._backbone)
    tap_t, saved = train_engine.forward(units, taps, x)
    acts, tape = saved
    fa = [_nchw(t) for t in tap_t]
    fc = mc._backbone_train_torch(x)
    ws = []
    for a, c in zip(fa, fc):
        assert a.shape == c.shape
        assert float((a - c.detach()).abs().max() / c.detach().abs().max()) < 3e-2 and _cos(a, c) > 0.9995
        ws.append(torch.randn_like(c) / c.numel() ** 0.5)
    for (k, ba), bc in zip(ma._backbone.named_buffers(), mc._backbone.buffers()):
        if ba.dtype == torch.int64:
            assert torch.equal(ba, bc), k
        else:
            torch.testing.assert_close(ba, bc, rtol=2e-2, atol=2e-3, msg=lambda m: k + ': ' + m)
    S = train_engine.LOSS_SCALE
    trace = []
    store = train_engine.backward(units, saved, {t: (w * S).permute(0, 2, 3, 1).contiguous().half() for t, w in zip(taps, ws)},
                                  trace=trace)
    assert len(trace) == len(units)
    _check_units_against_autograd(units, acts, tape, trace, S)
    # (c) end to end vs fp32 autograd of the modules (independent ReLU decisions: loose by nature)
    sum((c * w).sum() for c, w in zip(fc, ws)).backward()
    for (k, pa), pc in zip(ma._backbone.named_parameters(), mc._backbone.parameters()):
        g = store.get(pa)
        assert g is not None and g.shape == pa.shape, k
        assert _cos(g, pc.grad) > 0.9 and 0.8 < float(g.norm() / pc.grad.norm()) < 1.25, k
    # (d) end to end vs fp32 autograd over the same graph WITH THE HIP PATH'S OWN ReLU MASKS: tight
    ref, leaves = _mask_replay_reference(units, acts, x)
    sum((ref[t] * w).sum() for t, w in zip(taps, ws)).backward()
    worst = (1.0, '')
    for k, pa in ma._backbone.named_parameters():
        g, r = store.get(pa), leaves[id(pa)].grad
        cs, ratio = _cos(g, r), float(g.norm() / r.norm())
        worst = min(worst, (cs, k))
        assert cs > 0.999 and 0.98 < ratio < 1.02, (k, cs, ratio)
    print('mask-replay end-to-end gradients %s: worst cos %.6f (%s)' % (name, worst[0], worst[1]))


@pytest.mark.parametrize('ncls,two_convs,with_scale', [(1, True, True), (45, True, True), (45, False, False), (3, True, False)])
def test_head_output_glue_kernels_vs_torch_ops(ncls, two_convs, with_scale):
    """lfd_head_out_split_f16 / lfd_head_out_grad_f16 (csrc/head_out.hip) against the PyTorch ops they replace: slices of the
    padded conv output -> fp32 (x Scale) into the level-concatenated tensors; gradients -> dy fp16 (bit-identical: the same
    fp32 products, then one rounding), bias / Scale gradients accumulated (fp64 sums: tighter than torch's fp32 ones)."""
    n, h, w, P, p0, S = 3, 13, 21, 1000, 317, 1024.0
    g = torch.Generator(device='cuda').manual_seed(ncls)
    y = (torch.randn((n, h, w, 64), generator=g, device='cuda') * 2).half()
    scale = torch.tensor(1.37, device='cuda') if with_scale else None
    segs = [dict(kind='cls', channels=ncls, row0=0, scale=None)]
    if two_convs:
        segs.append(dict(kind='reg', channels=4, row0=ncls, scale=scale))
    outs = [torch.full((n, P, sg['channels']), -7.0, device='cuda') for sg in segs]
    ops.head_out_split(y, segs, outs, p0)
    yv = y.view(n, h * w, 64)
    for sg, o in zip(segs, outs):
        ref = yv[..., sg['row0']:sg['row0'] + sg['channels']].float()
        if sg['scale'] is not None:
            ref = ref * sg['scale']
        assert torch.equal(o[:, p0:p0 + h * w], ref)
        assert bool((o[:, :p0] == -7).all()) and bool((o[:, p0 + h * w:] == -7).all())      # nothing outside the level
    grads = [torch.randn((n, P, sg['channels']), generator=g, device='cuda') * 1e-3 for sg in segs]
    ref_dy = torch.zeros((n, h * w, 64), dtype=torch.float16, device='cuda')
    for sg, gr in zip(segs, grads):
        sg['dbias'] = torch.full((sg['channels'],), 0.5, device='cuda')
        sg['dscale'] = torch.full((), 0.25, device='cuda') if sg['scale'] is not None else None
        d = gr[:, p0:p0 + h * w]
        raw = yv[..., sg['row0']:sg['row0'] + sg['channels']].float()
        sg['ref_dscale'] = 0.25 + (d.double() * raw.double()).sum() if sg['scale'] is not None else None
        if sg['scale'] is not None:
            d = d * sg['scale']
        sg['ref_dbias'] = 0.5 + d.double().sum((0, 1))
        ref_dy[..., sg['row0']:sg['row0'] + sg['channels']] = (d * S).half()
    dy = ops.head_out_grad(y, segs, grads, p0, S)
    assert torch.equal(dy.view(n, h * w, 64), ref_dy)
    for sg in segs:
        torch.testing.assert_close(sg['dbias'].double(), sg['ref_dbias'], rtol=1e-6, atol=1e-7)
        if sg['scale'] is not None:
            torch.testing.assert_close(sg['dscale'].double(), sg['ref_dscale'], rtol=1e-6, atol=1e-7)
    dy2 = ops.head_out_grad(y, segs, grads, p0, S)           # deterministic; accumulates
    assert torch.equal(dy2, dy)


@pytest.mark.parametrize('name,hw', [('WIDERFACE_LFD_S', (160, 192)), ('TT100K_LFD_L', (128, 160)), ('WIDERFACE_LFD_XS', (96, 128))])
def test_whole_network_train_forward_backward(name, hw, monkeypatch):
    """LFD.forward in train mode = ONE autograd node on the HIP kernels (backbone, neck, GroupNorm towers shared by the
    levels, output convs + Scale): outputs vs the all-PyTorch forward; every unit of the backward vs autograd given the
    stored tensors; the output convs vs autograd; parameter gradients end to end (loose, see the backbone test)."""
    torch.manual_seed(3)
    ma = configs.build_model(name).cuda().train()
    configs.perturb_weights(ma)
    mc = copy.deepcopy(ma)
    x = torch.randn(3, 3, hw[0], hw[1], device='cuda')
    assert train_engine.network_supported(ma)
    monkeypatch.setenv('LFD_HIP_TRAIN', '1')
    ca, ra = ma(x)
    sizes_a = dict(ma._head_indexes_to_feature_map_sizes)
    monkeypatch.setenv('LFD_HIP_TRAIN', '0')
    cc, rc = mc(x)
    assert sizes_a == dict(mc._head_indexes_to_feature_map_sizes)
    assert ca.shape == cc.shape and ra.shape == rc.shape and ca.dtype == torch.float32
    assert _cos(ca, cc) > 0.999 and _cos(ra, rc) > 0.999
    assert _rel(ca, cc) < 3e-2 and _rel(ra, rc) < 3e-2
    wc, wr = torch.randn_like(cc) / cc.numel() ** 0.5, torch.randn_like(rc) / rc.numel() ** 0.5
    ((ca * wc).sum() + (ra * wr).sum()).backward()
    ((cc * wc).sum() + (rc * wr).sum()).backward()
    for (k, pa), pc in zip(ma.named_parameters(), mc.parameters()):
        assert pa.grad is not None and pa.grad.shape == pc.grad.shape, k
        assert _cos(pa.grad, pc.grad) > 0.9 and 0.8 < float(pa.grad.norm() / pc.grad.norm()) < 1.25, k
    # per-unit check of the same backward, driven by hand
    mb = copy.deepcopy(mc)
    mb.zero_grad()
    units, outs = train_engine.build_network(mb)
    _, saved = train_engine.forward(units, [], x)
    acts, tape = saved
    cls, reg, sizes, osaved = train_engine.outputs_forward(outs, acts, mb._num_heads)
    S = train_engine.LOSS_SCALE
    store = train_engine._GradStore()
    grads = train_engine.outputs_backward(outs, acts, osaved, sizes, wc, wr, store)
    starts = np.cumsum([0] + [h * w for h, w in sizes])
    for o in outs:                                         # output convs vs autograd
        xin = _nchw(acts[o.src]).requires_grad_(True)
        n, _, h, w = xin.shape
        lo, hi = starts[o.level], starts[o.level + 1]
        tot = 0
        for kind, conv in o.convs:
            wt, bs = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
            out = F.conv2d(xin, wt, bs).permute(0, 2, 3, 1).reshape(n, h * w, -1)
            if kind == 'reg' and o.scale is not None:
                out = out * o.scale._scale.detach()
            d = (wc if kind == 'cls' else wr)[:, lo:hi]
            assert _rel((cls if kind == 'cls' else reg)[:, lo:hi], out) < 2e-3
            tot = tot + (out * d).sum()
        tot.backward()
        assert _rel(_nchw(grads[o.src]) / S, xin.grad) < 3e-3
    trace = []
    train_engine.backward(units, saved, grads, store=store, trace=trace)
    assert len(trace) == len(units)
    _check_units_against_autograd(units, acts, tape, trace, S)
    # hand-driven serial unit schedule == the autograd node (network_backward: one batched final launch, one rounding per
    # parameter where the serial schedule rounds once per pyramid level): <= 1e-6 of the tensor's largest entry.  The stem's
    # parameters: <= 1e-4 (measured 4.6e-5, a BatchNorm weight) -- in the node the BatchNorm backward sums of the first unit of each stem pair come out of the 1x1
    # data-gradient conv's epilogue (another fp32 summation order -> the per-channel means of dy move by ~1e-7 relative -> a few
    # of its fp16 values by one ulp); everything behind the stem in the backward is untouched by that
    for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        tol = 1e-4 if k.startswith('_backbone._stem') else 1e-6
        assert float((pa.grad - store.get(pb)).abs().max()) <= tol * float(pa.grad.abs().max()) + 1e-30, k
    # end to end vs fp32 autograd over the same unit graph with the HIP path's own ReLU masks (see _mask_replay_reference):
    # backbone, neck, the GroupNorm towers shared by the pyramid levels, output convs and Scale -- tight
    ref, leaves = _mask_replay_reference(units, acts, x)
    tot = 0
    for o in outs:
        xin = ref[o.src]
        n, _, h, w = xin.shape
        lo, hi = starts[o.level], starts[o.level + 1]
        for kind, conv in o.convs:
            wl = leaves.setdefault(id(conv.weight), conv.weight.detach().clone().requires_grad_(True))
            bl = leaves.setdefault(id(conv.bias), conv.bias.detach().clone().requires_grad_(True))
            out = F.conv2d(xin, wl, bl).permute(0, 2, 3, 1).reshape(n, h * w, -1)
            if kind == 'reg' and o.scale is not None:
                sl = leaves.setdefault(id(o.scale._scale), o.scale._scale.detach().clone().requires_grad_(True))
                out = out * sl
            tot = tot + (out * (wc if kind == 'cls' else wr)[:, lo:hi]).sum()
    tot.backward()
    worst = (1.0, '')
    for k, pb in mb.named_parameters():
        g, r = store.get(pb), leaves[id(pb)].grad
        cs, ratio = _cos(g, r), float(g.norm() / (r.norm() + 1e-30))
        if pb.numel() == 1:
            # Scale (lfd_head.py: one scalar per level): its gradient is ONE sum of random-sign terms d * raw over the level's
            # regression outputs -- |sum| << sum |terms| under the test's random upstream gradient, so the fp16 rounding noise
            # of the terms does not cancel like the signal does (measured ratios 1.02 .. 1.09): sign + 15 %
            assert cs > 0 and 0.85 < ratio < 1.15, (k, cs, ratio)
            continue
        worst = min(worst, (cs, k))
        assert cs > 0.999 and 0.98 < ratio < 1.02, (k, cs, ratio)
    print('mask-replay end-to-end gradients %s: worst cos %.6f (%s)' % (name, worst[0], worst[1]))


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 2, 2), (2, 5, 7), (1, 16, 32), (1, 17, 33), (3, 31, 30), (2, 64, 64), (1, 160, 160), (4, 159, 161)])
@pytest.mark.parametrize('with_res', [False, True])
def test_stride2_data_gradient_per_parity_is_bit_identical_to_the_zero_inserted_conv(shape, with_res):
    """lfd_conv3x3s2_dgrad_nhwc_f16 (csrc/dgrad_s2.hip): dx per output parity (1 / 2 / 2 / 4 taps) == conv3x3 over the
    zero-inserted dy with the same data-gradient filter pack, bit for bit (the skipped products are exact zeros and the taps that
    remain are visited in the same order); also against a float64 conv_transpose of the fp16 operands."""
    import torch.nn.functional as F
    n, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + int(with_res))
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dy = (torch.randn(n, ho, wo, 64, generator=g) * 0.5).half().cuda()
    weight = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    res = (torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda() if with_res else None
    wp = ops.pack_conv_weight_train(weight, data_gradient=True)
    zeros = torch.zeros(64, device='cuda')
    ref = ops.conv2d_nhwc(ops.zero_insert2(dy, h, w), wp, zeros, 64, 64, 3, 1, False, residual=res)
    got = ops.conv3x3s2_dgrad(dy, wp, h, w, residual=res)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.isfinite(got.float()).all()
    bad = got != ref
    assert not bool(bad.any()), 'mismatches: %d of %d, first at %s' % (int(bad.sum()), bad.numel(), bad.nonzero()[0].tolist())
    # independent check: autograd's transposed convolution in float64 on the fp16-rounded operands
    w16 = weight.half().double().cpu()
    dy64 = dy.double().cpu().permute(0, 3, 1, 2)
    dx64 = F.conv_transpose2d(dy64, w16, stride=2, padding=1, output_padding=(h - 1 - 2 * (ho - 1), w - 1 - 2 * (wo - 1)))
    if with_res:
        dx64 = dx64 + res.double().cpu().permute(0, 3, 1, 2)
    err = (got.double().cpu().permute(0, 3, 1, 2) - dx64).abs()
    assert bool((err <= 2e-3 * dx64.abs().clamp(min=1.0)).all()), float(err.max())


@pytest.mark.parametrize('name,shape', [('WIDERFACE_LFD_S', (2, 128, 160)), ('TT100K_LFD_L', (2, 96, 128)), ('WIDERFACE_LFD_XS', (3, 192, 224)),
                                        ('WIDERFACE_LFD_S', (3, 100, 132))])
def test_network_schedules_equal_the_serial_unit_schedule(name, shape):
    """train_engine.network_forward / network_backward (round 4) against the serial unit API (train_engine.forward /
    outputs_forward / outputs_backward / backward: the launches in list order, a final launch per conv), in both forms:
      level by level          weight gradients into private partial buffers + ONE batched final launch: outputs and BatchNorm
                              statistics bit for bit; parameter gradients equal up to the ONE rounding the batched final saves
                              per shared parameter (<= 1e-6 of the tensor's largest entry); a second run (persistent buffers
                              re-used) bit for bit equal to the first;
      level-concatenated      (CONCAT_HEAD, the default: shared towers once over all levels) the same network up to the order
                              of GroupNorm's partial sums (another block partition of a level's pixels): outputs within 2e-3
                              of the logit scale, gradients cos > 0.9999 and norm within 0.2 % per parameter (measured: cos
                              1.000000, norm 7e-8)."""
    import copy
    from lfd_amd import configs, train_engine
    torch.manual_seed(3)
    m0 = configs.build_model(name).cuda().train()
    configs.perturb_weights(m0, seed=1)
    n, h, w = shape
    x = (torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()

    def serial():
        m = copy.deepcopy(m0)
        units, outs = train_engine.build_network(m)
        _, saved = train_engine.forward(units, [], x)
        cls, reg, sizes, osaved = train_engine.outputs_forward(outs, saved[0], m._num_heads)
        g = torch.Generator(device='cuda').manual_seed(9)
        wc, wr = torch.randn(cls.shape, generator=g, device='cuda') * 1e-2, torch.randn(reg.shape, generator=g, device='cuda') * 1e-2
        store = train_engine._GradStore()
        grads = train_engine.outputs_backward(outs, saved[0], osaved, sizes, wc, wr, store)
        train_engine.backward(units, saved, grads, store=store)
        return m, cls, reg, wc, wr, {k: store.get(p) for k, p in m.named_parameters()}

    def node(concat):
        m = copy.deepcopy(m0)
        keep = train_engine.CONCAT_HEAD
        train_engine.CONCAT_HEAD = concat
        try:
            cls, reg = m(x)
            torch.autograd.backward([cls, reg], [wc, wr])
            torch.cuda.synchronize()
        finally:
            train_engine.CONCAT_HEAD = keep
        return m, cls.detach(), reg.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}

    ms, cls_s, reg_s, wc, wr, gs = serial()
    res = []
    for rep in range(2):                                   # twice: the persistent partial buffers are re-used
        mp, cls_p, reg_p, gp = node(False)
        assert torch.equal(cls_p, cls_s) and torch.equal(reg_p, reg_s)
        for (k, a), b in zip(ms.state_dict().items(), mp.state_dict().values()):
            assert torch.equal(a, b), k                    # BatchNorm running statistics, num_batches_tracked
        worst = 0.0
        for k in gs:
            a, b = gs[k], gp[k]
            e = float((a - b).abs().max() / a.abs().max().clamp_min(1e-20))
            worst = max(worst, e)
            # (the stem: the node takes the BatchNorm backward sums of the first unit of each pair in the 1x1 data-gradient conv's
            #  epilogue -- another fp32 summation order, a few fp16 values of dy one ulp apart -- see the whole-network test)
            assert e <= (1e-4 if k.startswith('_backbone._stem') else 1e-6), (k, e)
        res.append(gp)
        print('%s level by level: worst relative gradient difference to the serial schedule %.2e' % (name, worst))
    for k in gs:
        assert torch.equal(res[0][k], res[1][k]), k
    # the level-concatenated form
    mc, cls_c, reg_c, gc = node(True)
    assert mc.__dict__.get('_lfd_concat_layout') is not None
    lim = 2e-3 * float(cls_s.abs().max().clamp_min(1.0))
    assert float((cls_c - cls_s).abs().max()) <= lim and float((reg_c - reg_s).abs().max()) <= lim
    worst_cos, worst_norm = 1.0, 0.0
    for k in gs:
        a, b = gs[k].double().reshape(-1), gc[k].double().reshape(-1)
        if float(a.norm()) == 0.0:
            assert float(b.norm()) == 0.0, k
            continue
        cos = float(a @ b / (a.norm() * b.norm()))
        nr = abs(float(b.norm() / a.norm()) - 1.0)
        worst_cos, worst_norm = min(worst_cos, cos), max(worst_norm, nr)
        assert cos > 0.9999 and nr < 2e-3, (k, cos, nr)
    for (k, a), b in zip(ms.state_dict().items(), mc.state_dict().values()):
        if 'running_' in k:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=k)
    print('%s level-concatenated: worst gradient cosine %.6f, worst norm deviation %.2e' % (name, worst_cos, worst_norm))


@pytest.mark.parametrize('nhw', [(2, 70, 94), (3, 128, 160)])
def test_first_unit_backward_without_a_dy_tensor_is_bit_identical(nhw):
    """lfd_stem_conv0_bn_bwd_wgrad (round 4): BatchNorm's sums and the first conv's weight gradient straight from dz and y -- the
    unit has no data gradient, so the apply pass that would write dy is folded into the weight gradient's loader -- against
    lfd_bn_train_bwd_f16 + lfd_stem_conv0_wgrad: dgamma, dbeta and dW bit for bit (the same arithmetic in the same order)."""
    n, h, w = nhw
    c = 64
    g = torch.Generator(device='cuda').manual_seed(21)
    x = torch.randn(n, 3, h, w, generator=g, device='cuda')
    wt = torch.randn(c, 3, 3, 3, generator=g, device='cuda') * 0.2
    gamma = torch.empty(c, device='cuda').uniform_(0.5, 1.5)
    beta = torch.empty(c, device='cuda').normal_(0, 0.3)
    y, stats = ops.stem_conv0_train_fwd_bn_stats(x, wt, 1e-5, 0.1)
    dz = (torch.randn(y.shape, generator=g, device='cuda') * 0.5).half()
    inv = 1.0 / 64
    dga, dba, dwa = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), torch.zeros_like(wt)
    dy, _ = ops.bn_train_backward(dz, y, None, stats, gamma, inv, dga, dba, want_g=False, accumulate=True, relu=True, beta=beta)
    ops.stem_conv0_wgrad(x, dy, inv, out=dwa, accumulate=True)
    dgb, dbb, dwb = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), torch.zeros_like(wt)
    ops.stem_conv0_bn_bwd_wgrad(x, dz, y, stats, gamma, beta, inv, dgb, dbb, dwb)
    assert torch.equal(dga, dgb) and torch.equal(dba, dbb)
    assert float(dwa.abs().max()) > 0 and torch.equal(dwa, dwb)


@pytest.mark.parametrize('nhw', [(2, 37, 53), (1, 8, 8), (3, 130, 70), (2, 160, 160)])
@pytest.mark.parametrize('cout', [64])
def test_conv1x1_of_an_unstored_bn_relu_activation_is_bit_identical(nhw, cout):
    """lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16 / lfd_conv1x1_wgrad_partials_of_bn_relu_f16 (round 4): the second conv of a stem
    pair normalises + ReLUs its operand inside the kernels, the producer's activation is never stored -- against
    lfd_bn_train_apply_f16 followed by the plain kernels: conv output, its batch statistics, the running statistics and the
    weight-gradient partial sums bit for bit."""
    n, h, w = nhw
    cin = 64
    g = torch.Generator(device='cuda').manual_seed(5 + h)
    yin = (torch.randn(n, h, w, cin, generator=g, device='cuda') * 1.3 + 0.2).half()
    pstats = ops.bn_train_stats(yin, 1e-5, 0.1, None, None)
    pg = torch.empty(cin, device='cuda').uniform_(0.5, 1.5)
    pb = torch.empty(cin, device='cuda').normal_(0, 0.3)
    wt = torch.randn(cout, cin, 1, 1, generator=g, device='cuda') * 0.15
    wp = ops.pack_conv_weight_train(wt)
    zb = torch.zeros(cout, device='cuda')
    rm_a, rv_a = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
    rm_b, rv_b = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
    z = ops.bn_train_apply(yin, pstats, pg, pb, None, True)
    ya, sa = ops.conv2d_bn_stats(z, wp, zb, cin, cout, 1, 1, 1e-5, 0.1, rm_a, rv_a)
    yb, sb = ops.conv1x1_of_bn_relu_bn_stats(yin, pstats, pg, pb, wp, zb, cout, 1e-5, 0.1, rm_b, rv_b)
    assert float(ya.float().abs().max()) > 0 and torch.equal(ya, yb)
    assert torch.equal(sa, sb) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    dy = (torch.randn(n, h, w, cout, generator=g, device='cuda') * 0.5).half()
    floats, nwg, nblk = ops.conv_wgrad_partial_floats(z, dy, 1, 1)
    pa, pb_ = torch.zeros(floats, device='cuda'), torch.zeros(floats, device='cuda')
    ops.conv_wgrad_partials(z, dy, 1, 1, pa)
    ops.conv1x1_wgrad_partials_of_bn_relu(yin, pstats, pg, pb, dy, pb_)
    assert float(pa.abs().max()) > 0 and torch.equal(pa, pb_)


def test_network_with_and_without_stored_stem_activations_is_bit_identical(monkeypatch):
    """LFD_BN_APPLY_IN_CONV=0 (every unit stores its activation) against the default (the first conv of each stem pair hands its
    pre-normalisation output to the 1x1 conv that follows): logits, loss-side gradients and every parameter gradient bit for
    bit; the two activations are really absent from the saved state."""
    from lfd_amd import configs, train_engine as TE
    res = []
    for flag in ('1', '0'):
        monkeypatch.setenv('LFD_BN_APPLY_IN_CONV', flag)
        torch.manual_seed(0)
        m = configs.build_model('WIDERFACE_LFD_S')
        configs.perturb_weights(m)
        m.train().cuda()
        units, outs = TE.build_network(m)
        plan = (units, outs, m._num_heads)
        d = TE._deferred_units(units, outs)
        assert (len(d) == 2 and all(units[v].conv.kernel_size[0] == 1 for v in d.values())) if flag == '1' else d == {}
        x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(1)).cuda()
        for p in m.parameters():
            p.grad = None
        cls, reg, sizes, saved = TE.network_forward(m, plan, x)
        (acts, tape), _, _ = saved
        assert [acts[units[u].dst] is None for u in d] == [True] * len(d)
        gc = torch.randn(cls.shape, generator=torch.Generator().manual_seed(2)).cuda() * 1e-2
        gr = torch.randn(reg.shape, generator=torch.Generator().manual_seed(3)).cuda() * 1e-2
        TE.network_backward(m, plan, saved, sizes, gc, gr, 64.0)
        torch.cuda.synchronize()
        res.append((cls.clone(), reg.clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
        del m
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert len(res[0][2]) == len(res[1][2]) > 50
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))


@pytest.mark.parametrize('nhw', [(2, 37, 53), (1, 8, 8), (3, 130, 70), (2, 160, 160)])
def test_batchnorm_backward_sums_in_the_data_gradient_convs_epilogue(nhw):
    """lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16 + lfd_bn_train_bwd_rows_f16 (round 4): the 1x1 data-gradient conv of a stem pair
    leaves the BatchNorm backward sums of the unit in front of it behind -- against the plain conv followed by lfd_bn_train_bwd_f16:
    dz bit for bit (the same conv), dgamma / dbeta to fp32 summation order, dy within an fp16 ulp."""
    n, h, w = nhw
    c = 64
    g = torch.Generator(device='cuda').manual_seed(9 + h)
    y_u = (torch.randn(n, h, w, c, generator=g, device='cuda') * 1.2 + 0.1).half()
    stats = ops.bn_train_stats(y_u, 1e-5, 0.1, None, None)
    gamma = torch.empty(c, device='cuda').uniform_(0.5, 1.5)
    beta = torch.empty(c, device='cuda').normal_(0, 0.3)
    dyv = (torch.randn(n, h, w, c, generator=g, device='cuda') * 0.5).half()
    wt = torch.randn(c, c, 1, 1, generator=g, device='cuda') * 0.15
    wp = ops.pack_conv_weight_train(wt, data_gradient=True)
    zb = torch.zeros(c, device='cuda')
    inv = 1.0 / 64
    dz_a = ops.conv2d_nhwc(dyv, wp, zb, c, c, 1, 1, False)
    dga, dba = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    dy_a, _ = ops.bn_train_backward(dz_a, y_u, None, stats, gamma, inv, dga, dba, want_g=False, accumulate=True, relu=True, beta=beta)
    dz_b, rows = ops.conv1x1_dgrad_bn_bwd_sums(dyv, wp, zb, y_u, stats, gamma, beta)
    dgb, dbb = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    dy_b = ops.bn_train_backward_rows(dz_b, y_u, stats, gamma, beta, inv, dgb, dbb, rows)
    torch.cuda.synchronize()
    assert rows >= 1 and float(dz_a.float().abs().max()) > 0 and torch.equal(dz_a, dz_b)
    assert _rel(dgb, dga) < 2e-6 and _rel(dbb, dba) < 2e-6
    tol = dy_a.float().abs() * 2.0 ** -10 + 2e-6 * float(dy_a.float().abs().max()) + 1e-7
    assert bool(((dy_a.float() - dy_b.float()).abs() <= tol).all())
    assert float((dy_a != dy_b).float().mean()) < 2e-3


def test_network_with_and_without_epilogue_sums_agree(monkeypatch):
    """LFD_BN_SUMS_IN_DGRAD=0 (every BatchNorm backward takes its own sums pass) against the default: the same logits, parameter
    gradients to fp32 summation order"""
    from lfd_amd import configs, train_engine as TE
    res = []
    for flag in ('1', '0'):
        monkeypatch.setenv('LFD_BN_SUMS_IN_DGRAD', flag)
        torch.manual_seed(0)
        m = configs.build_model('WIDERFACE_LFD_S')
        configs.perturb_weights(m)
        m.train().cuda()
        units, outs = TE.build_network(m)
        plan = (units, outs, m._num_heads)
        x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(1)).cuda()
        for p in m.parameters():
            p.grad = None
        cls, reg, sizes, saved = TE.network_forward(m, plan, x)
        gc = torch.randn(cls.shape, generator=torch.Generator().manual_seed(2)).cuda() * 1e-2
        gr = torch.randn(reg.shape, generator=torch.Generator().manual_seed(3)).cuda() * 1e-2
        TE.network_backward(m, plan, saved, sizes, gc, gr, 64.0)
        torch.cuda.synchronize()
        res.append((cls.clone(), [(n_, p.grad.clone()) for n_, p in m.named_parameters() if p.grad is not None]))
        del m
    assert torch.equal(res[0][0], res[1][0])
    worst = max((_rel(a, b), n_) for (n_, a), (_, b) in zip(res[0][1], res[1][1]))
    assert worst[0] < 2e-5, worst


@pytest.mark.parametrize('ncls', [1, 45])
def test_head_output_glue_of_all_levels_in_one_launch_is_bit_identical(ncls):
    """lfd_head_out_split_levels_f16 / lfd_head_out_grad_levels_f16 (round 4): the per-level calls of the `_concat` entry points for
    all pyramid levels of one output conv as ONE launch each (levels = blockIdx.y, per-level Scale) -- outputs, dy, the shared
    bias gradients and the per-level Scale gradients bit for bit those of the call sequence."""
    n, S = 3, 1024.0
    hws = [80 * 80, 40 * 40, 20 * 20, 10 * 10, 5 * 5]
    starts = [0]
    for hw in hws:
        starts.append(starts[-1] + hw)
    P = starts[-1]
    g = torch.Generator(device='cuda').manual_seed(7 + ncls)
    y = (torch.randn((n, P, 64), generator=g, device='cuda') * 2).half()
    scales = [torch.tensor(0.7 + 0.2 * l, device='cuda') for l in range(len(hws))]
    grads = [torch.randn((n, P, c), generator=g, device='cuda') * 1e-3 for c in (ncls, 4)]

    def run(batched):
        outs = [torch.full((n, P, c), -7.0, device='cuda') for c in (ncls, 4)]
        dbias = [torch.full((c,), 0.5, device='cuda') for c in (ncls, 4)]
        dscale = [torch.full((), 0.25, device='cuda') for _ in hws]
        dy = torch.full((n, P, 64), 3.0, dtype=torch.float16, device='cuda')
        lv_f, lv_b = [], []
        for l, hw in enumerate(hws):
            segs = [dict(kind='cls', channels=ncls, row0=0, scale=None, dbias=dbias[0], dscale=None),
                    dict(kind='reg', channels=4, row0=ncls, scale=scales[l], dbias=dbias[1], dscale=dscale[l])]
            lv_f.append((hw, starts[l], segs, outs))
            lv_b.append((hw, starts[l], segs, grads))
        if batched:
            ops.head_out_split_levels(y, lv_f)
            ops.head_out_grad_levels(y, lv_b, S, dy)
        else:
            for hw, p0, segs, o in lv_f:
                ops.head_out_split_concat(y, hw, segs, o, p0)
            for hw, p0, segs, gr in lv_b:
                ops.head_out_grad_concat(y, hw, segs, gr, p0, S, dy)
        torch.cuda.synchronize()
        return outs + [dy] + dbias + dscale

    a, b = run(True), run(False)
    assert float(a[0].abs().max()) > 0 and all(torch.equal(x, z) for x, z in zip(a, b))
    assert not bool((a[2] == 3.0).all())


def test_batchnorm_backward_of_all_neck_levels_in_three_launches_is_bit_identical():
    """lfd_bn_train_bwd_from_levels_f16 (round 4) against lfd_bn_train_bwd_from_f16 level by level: dy, dgamma, dbeta bit for bit"""
    n, c = 3, 128
    shapes = [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)]
    starts = [0]
    for h, w in shapes:
        starts.append(starts[-1] + h * w)
    P = starts[-1]
    g = torch.Generator(device='cuda').manual_seed(23)
    dz = (torch.randn(n, P, c, generator=g, device='cuda') * 0.5).half()
    ys = [(torch.randn(n, h, w, c, generator=g, device='cuda') * 1.3 + 0.2).half() for h, w in shapes]
    stats = [ops.bn_train_stats(y, 1e-5, 0.1, None, None) for y in ys]
    gam = [torch.empty(c, device='cuda').uniform_(0.5, 1.5) for _ in shapes]
    bet = [torch.empty(c, device='cuda').normal_(0, 0.3) for _ in shapes]
    inv = 1.0 / 64

    def run(batched):
        dg = [torch.full((c,), 0.5, device='cuda') for _ in shapes]
        db = [torch.full((c,), -0.25, device='cuda') for _ in shapes]
        if batched:
            dys = ops.bn_train_backward_from_levels(dz, [(starts[l], ys[l], stats[l], gam[l], bet[l], dg[l], db[l])
                                                         for l in range(len(shapes))], inv)
        else:
            dys = [ops.bn_train_backward_from(dz, starts[l], ys[l], stats[l], gam[l], bet[l], inv, dg[l], db[l]) for l in range(len(shapes))]
        torch.cuda.synchronize()
        return dys + dg + db

    a, b = run(True), run(False)
    assert float(a[0].float().abs().max()) > 0 and all(torch.equal(x, z) for x, z in zip(a, b))


def test_network_with_levels_batched_and_level_by_level_is_bit_identical(monkeypatch):
    """LFD_BN_LEVELS / LFD_HEAD_OUT_LEVELS = 0 (the neck units' BatchNorm passes and the output convs' glue one pyramid level after
    the other) against the default (all levels per launch): logits, running statistics and every parameter gradient bit for bit"""
    from lfd_amd import configs, train_engine as TE
    res = []
    for flag in ('1', '0'):
        monkeypatch.setenv('LFD_BN_LEVELS', flag)
        monkeypatch.setenv('LFD_HEAD_OUT_LEVELS', flag)
        torch.manual_seed(0)
        m = configs.build_model('WIDERFACE_LFD_S')
        configs.perturb_weights(m)
        m.train().cuda()
        units, outs = TE.build_network(m)
        plan = (units, outs, m._num_heads)
        x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(1)).cuda()
        for p in m.parameters():
            p.grad = None
        cls, reg, sizes, saved = TE.network_forward(m, plan, x)
        assert isinstance(saved[1], dict)                  # the level-concatenated schedule ran
        gc = torch.randn(cls.shape, generator=torch.Generator().manual_seed(2)).cuda() * 1e-2
        gr = torch.randn(reg.shape, generator=torch.Generator().manual_seed(3)).cuda() * 1e-2
        TE.network_backward(m, plan, saved, sizes, gc, gr, 64.0)
        torch.cuda.synchronize()
        res.append((cls.clone(), reg.clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None],
                    [b.clone() for b in m.buffers()]))
        del m
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2])) and len(res[0][2]) > 50
    assert all(torch.equal(a, b) for a, b in zip(res[0][3], res[1][3]))
