"""Training-iteration device path (SURVEY 8a rows 14 + 18): flat-buffer SGD + gradient clipping kernels
(csrc/optim.hip) against torch.optim.SGD / torch clip_grad_norm_ running on the same GPU, and one whole
train_step (forward, fused get_loss, backward, clip, update) against the op-by-op composition."""
import copy

import numpy as np
import pytest
import torch

from lfd_amd import configs, optim, train

pytestmark = pytest.mark.gpu


def _pair(name='WIDERFACE_LFD_XS'):
    a = configs.build_model(name).cuda()
    b = copy.deepcopy(a)
    return a, b


def _set_grads(ma, mb, seed, scale):
    g = torch.Generator(device='cuda').manual_seed(seed)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        r = torch.randn(pa.shape, generator=g, device='cuda') * scale
        pa.grad.copy_(r)
        pb.grad = r.clone()


@pytest.mark.parametrize('kw', [dict(lr=0.1, momentum=0.9, weight_decay=1e-4),
                                dict(lr=0.05, momentum=0.0, weight_decay=0.0),
                                dict(lr=0.02, momentum=0.8, dampening=0.1, weight_decay=5e-4),
                                dict(lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-4)])
def test_flat_sgd_with_clipping_matches_torch(kw):
    ma, mb = _pair()
    oa = optim.SGD(ma.parameters(), **kw)
    ob = torch.optim.SGD(mb.parameters(), **kw)
    oa.zero_grad()
    for it in range(4):
        scale = [1.0, 0.001, 3.0, 0.1][it]                     # norms above and below max_norm
        _set_grads(ma, mb, 100 + it, scale)
        if it == 3:                                            # plain step, no clipping
            oa.step()
            ob.step()
        else:
            na = oa.clip_and_step(10.0)
            nb = torch.nn.utils.clip_grad_norm_(list(mb.parameters()), max_norm=10.0, norm_type=2)
            ob.step()
            assert float(na) == pytest.approx(float(nb), rel=2e-6)
            for pa, pb in zip(ma.parameters(), mb.parameters()):   # clip_grad_norm_ leaves the scaled gradients behind
                torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-6, atol=1e-12)
        for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            torch.testing.assert_close(pa, pb, rtol=3e-6, atol=2e-8, msg=lambda m: '%s step %d: %s' % (k, it, m))
        if kw['momentum']:
            for pa, pb in zip(ma.parameters(), mb.parameters()):
                torch.testing.assert_close(oa.state[pa]['momentum_buffer'], ob.state[pb]['momentum_buffer'],
                                           rtol=3e-6, atol=2e-8)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa['param_groups'][0]['params'] == sb['param_groups'][0]['params']
    assert set(sa['state']) == set(sb['state'])


def test_standalone_clip_grad_norm_matches_torch():
    ma, mb = _pair()
    oa = optim.SGD(ma.parameters(), lr=0.1)
    oa.zero_grad()
    for scale in (2.0, 1e-4):
        _set_grads(ma, mb, 7, scale)
        na = optim.clip_grad_norm_(ma.parameters(), max_norm=10, norm_type=2)
        nb = torch.nn.utils.clip_grad_norm_(list(mb.parameters()), max_norm=10, norm_type=2)
        assert float(na) == pytest.approx(float(nb), rel=2e-6)
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            torch.testing.assert_close(pa.grad, pb.grad, rtol=2e-6, atol=1e-12)
    with pytest.raises(NotImplementedError):
        optim.clip_grad_norm_(ma.parameters(), max_norm=10, norm_type=1)
    with pytest.raises(RuntimeError):
        optim.clip_grad_norm_(list(mb.parameters()), max_norm=10)      # not owned by a flat optimizer


def test_state_dict_round_trip_and_gradient_adoption():
    ma, mb = _pair()
    oa = optim.SGD(ma.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    oa.zero_grad()
    _set_grads(ma, mb, 1, 1.0)
    oa.step()
    sd = copy.deepcopy(oa.state_dict())
    mb.load_state_dict(ma.state_dict())
    ob = optim.SGD(mb.parameters(), lr=0.5, momentum=0.1)
    ob.load_state_dict(sd)
    assert ob.param_groups[0]['lr'] == 0.1 and ob.param_groups[0]['momentum'] == 0.9
    for p in mb.parameters():                       # a foreign zero_grad(set_to_none=True) + fresh .grad tensors
        p.grad = None
    g = torch.Generator(device='cuda').manual_seed(3)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        r = torch.randn(pa.shape, generator=g, device='cuda')
        pa.grad.copy_(r)
        pb.grad = r.clone()
    oa.step()
    ob.step()
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa, pb)
        assert torch.equal(oa.state[pa]['momentum_buffer'], ob.state[pb]['momentum_buffer'])


def _annotations(rng, n, hw):
    ann = []
    for _ in range(n):
        k = 5
        wh = np.exp(rng.uniform(np.log(8), np.log(90), (k, 2)))
        xy = rng.uniform(0, 1, (k, 2)) * (np.array([hw[1], hw[0]]) - wh).clip(1)
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), np.zeros(k, np.int64)))
    return ann


def test_train_step_matches_op_by_op_composition(monkeypatch):
    """3 iterations of train_step (fused get_loss + flat SGD with fused clipping) vs the same iterations done with the
    op-by-op get_loss mirror, torch clip_grad_norm_ and torch.optim.SGD.  The convolution backward runs through
    PyTorch-ROCm in both (SURVEY row 18 is only partly hand-written), so the comparison isolates the new kernels."""
    rng = np.random.default_rng(11)
    ma, mb = _pair()
    ma.train()
    mb.train()
    kw = dict(lr=0.01, momentum=0.9, weight_decay=1e-4)
    oa, ob = optim.SGD(ma.parameters(), **kw), torch.optim.SGD(mb.parameters(), **kw)
    clip = dict(max_norm=10, norm_type=2)
    for it in range(3):
        # both sides start every iteration from identical parameters / BN statistics / momentum, so that the comparison
        # is per update and the (atomics-order) noise of the PyTorch-ROCm backward does not compound
        mb.load_state_dict(ma.state_dict())
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            if 'momentum_buffer' in oa.state.get(pa, {}):
                ob.state[pb]['momentum_buffer'].copy_(oa.state[pa]['momentum_buffer'])
        start = [p.detach().clone() for p in mb.parameters()]
        x = torch.from_numpy(rng.normal(0, 1, (4, 3, 160, 192)).astype(np.float32)).cuda()
        ann = _annotations(rng, 4, (160, 192))
        monkeypatch.setenv('LFD_FUSED_LOSS', '1')
        lva, na = train.train_step(ma, oa, x, ann, clip, clip_active=True)
        monkeypatch.setenv('LFD_FUSED_LOSS', '0')
        out = mb.get_loss(mb(x), ann)
        ob.zero_grad()
        out['loss'].backward()
        nb = torch.nn.utils.clip_grad_norm_(list(mb.parameters()), **clip)
        ob.step()
        assert lva['loss'] == pytest.approx(out['loss_values']['loss'], rel=1e-4), it
        assert lva['regression_loss'] > 0
        assert float(na) == pytest.approx(float(nb), rel=1e-3)
        for (k, pa), pb, p0 in zip(ma.named_parameters(), mb.parameters(), start):
            upd = float((pb.detach() - p0).abs().max())
            assert float((pa.detach() - pb.detach()).abs().max()) <= 2e-3 * upd + 2.5e-7, (it, k, upd)      # + 2 ulp at 1.0 (norm weights)


def test_inference_plan_sees_the_updated_parameters():
    """The update kernel writes the parameters behind autograd's back; the version counters are bumped so that the
    folded fp16 inference plan (engine.get_plan) is rebuilt."""
    m = configs.build_model('WIDERFACE_LFD_XS').cuda()
    configs.perturb_weights(m)
    o = optim.SGD(m.parameters(), lr=0.5, momentum=0.9)
    x = torch.randn(1, 3, 128, 160, device='cuda')
    m.eval()
    with torch.no_grad():
        c0, _ = m(x)
    o.zero_grad()
    g = torch.Generator(device='cuda').manual_seed(0)
    for p in m.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g, device='cuda') * 0.05)
    o.step()
    with torch.no_grad():
        c1, _ = m(x)
    assert not torch.equal(c0, c1)
    fresh = configs.build_model('WIDERFACE_LFD_XS').cuda()
    fresh.load_state_dict(m.state_dict())
    fresh.eval()
    with torch.no_grad():
        c2, _ = fresh(x)
    assert torch.equal(c1, c2)


def test_training_on_a_fixed_batch_reduces_the_loss_like_the_pytorch_path(monkeypatch):
    """40 iterations on one fixed synthetic batch (WIDERFACE_LFD_XS, 8 x 256x256): the all-HIP iteration (whole-network
    forward + backward kernels, fused get_loss, clip + flat SGD) must drive the loss down, and about as far as the same
    modules do through PyTorch-ROCm autograd + op-by-op loss + torch.optim.SGD from the same initial weights."""
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.normal(0, 1, (8, 3, 256, 256)).astype(np.float32)).cuda()
    ann = _annotations(rng, 8, (256, 256))
    kw = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    clip = dict(max_norm=10, norm_type=2)
    curves = {}
    for mode in ('hip', 'torch'):
        monkeypatch.setenv('LFD_HIP_TRAIN', '1' if mode == 'hip' else '0')
        monkeypatch.setenv('LFD_FUSED_LOSS', '1' if mode == 'hip' else '0')
        torch.manual_seed(5)
        m = configs.build_model('WIDERFACE_LFD_XS').cuda().train()
        opt = optim.SGD(m.parameters(), **kw) if mode == 'hip' else torch.optim.SGD(m.parameters(), **kw)
        losses = []
        for it in range(40):
            lv, gn = train.train_step(m, opt, x, ann, clip, clip_active=True)
            assert np.isfinite(lv['loss']) and np.isfinite(float(gn)), (mode, it)
            losses.append(lv['loss'])
        curves[mode] = losses
    h, t = curves['hip'], curves['torch']
    assert h[0] == pytest.approx(t[0], rel=2e-2)                     # same start (fp16 vs fp32 forward)
    assert np.mean(h[-5:]) < 0.6 * h[0], h                           # it learns
    assert np.mean(h[-5:]) < 1.25 * np.mean(t[-5:]) + 0.05, (h[-5:], t[-5:])
