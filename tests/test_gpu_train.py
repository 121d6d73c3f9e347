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


def test_graphed_train_step_equals_the_eager_iterations():
    """GraphedTrainStep: the whole iteration (forward, device targets, fused loss, hand-written backward, clip + SGD) replayed
    as ONE HIP graph -- same losses, gradient norms, parameters, BatchNorm statistics and momentum buffers as train_step,
    bit for bit (the kernels are deterministic), over changing batches and annotation counts; a changed learning rate runs
    eagerly once and is captured again; the inference plan sees the replayed updates."""
    rng = np.random.default_rng(3)
    torch.manual_seed(5)
    ma = configs.build_model('WIDERFACE_LFD_XS').cuda().train()
    mb = configs.build_model('WIDERFACE_LFD_XS').cuda().train()
    mb.load_state_dict(ma.state_dict())
    kw = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    oa, ob = optim.SGD(ma.parameters(), **kw), optim.SGD(mb.parameters(), **kw)
    clip = dict(max_norm=10, norm_type=2)
    step = train.GraphedTrainStep(mb, ob, clip, max_boxes=64)
    xe = torch.randn(1, 3, 160, 192, device='cuda')
    mb.eval()
    with torch.no_grad():
        c0, _ = mb(xe)
    mb.train()
    replays = 0
    for it in range(7):
        if it == 5:                                   # a learning-rate change: new graph key
            for o in (oa, ob):
                o.param_groups[0]['lr'] = 0.01
        x = torch.from_numpy(rng.normal(0, 1, (4, 3, 160, 192)).astype(np.float32)).cuda()
        ann = _annotations(rng, 4, (160, 192))
        if it % 2:
            ann[1] = (ann[1][0][:2], ann[1][1][:2])   # varying annotation counts, incl. an image without boxes
            ann[2] = (ann[2][0][:0], ann[2][1][:0])
        lva, na = train.train_step(ma, oa, x, ann, clip, True)
        lvb, nb = step(x, ann, True)
        replays += len(step.graphs) > 0
        assert lva == lvb, (it, lva, lvb)
        assert float(na) == float(nb), it
        for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            assert torch.equal(pa, pb), (it, k)
        for (k, ba), bb in zip(ma.named_buffers(), mb.buffers()):
            assert torch.equal(ba, bb), (it, k)
    assert len(step.graphs) == 2 and replays >= 5      # iterations 1-4 and 6 were graph replays
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(oa.state[pa]['momentum_buffer'], ob.state[pb]['momentum_buffer'])
    mb.eval()
    with torch.no_grad():
        c1, _ = mb(xe)
    assert not torch.equal(c0, c1)                     # the engine re-packed the weights the replays wrote
    mb.train()
    # sync=False (round 4): five iterations enqueued without reading anything back -- the host runs ahead through the rings
    # of pinned annotation / loss buffers -- give the same values and parameters as the synchronous calls on the twin
    batches = [(torch.from_numpy(rng.normal(0, 1, (4, 3, 160, 192)).astype(np.float32)).cuda(), _annotations(rng, 4, (160, 192)))
               for _ in range(5)]
    pend = [step(x, ann, True, sync=False)[0] for x, ann in batches]
    want = [train.train_step(ma, oa, x, ann, clip, True)[0] for x, ann in batches]
    assert pend[-1].get() == want[-1] and pend[-2].get() == want[-2] and pend[-3].get() == want[-3]
    torch.cuda.synchronize()
    for (k, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(pa, pb), k
    with pytest.raises(RuntimeError):
        step(torch.randn(2, 3, 160, 192, device='cuda'), _annotations(rng, 2, (160, 192)))     # another batch shape


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


def test_config5_widerface_s_640_loss_curve_vs_fp32_autograd(monkeypatch):
    """BASELINE config 5 protocol on one GPU's share: WIDERFACE_LFD_S from scratch (reference init, seed of the config),
    synthetic 640x640 frames, G ~ U{1..20} boxes per image with w, h ~ logU[6, 300], a FRESH batch of 32 (the
    config's per-GPU batch) every iteration, SGD momentum 0.9 / weight decay 1e-4 / lr 0.1 with the config's linear warm-up
    (ratio 0.1 over 200 iterations, WIDERFACE_LFD_S.py:218-241), clip_grad_norm_(10) -- 40 iterations.  The all-HIP
    iteration (fp16 activations, dynamic loss scale starting at 1024, fp32 accumulate / parameters) against an INDEPENDENT
    fp32 path: this package's mirror nn.Modules (same parameters, same semantics as the reference's -- pinned by the
    reference-generated goldens) through PyTorch-ROCm autograd (MIOpen fp32 convolutions) + the op-by-op loss +
    torch.optim.SGD, from identical initial weights.  The loss CURVES must agree: 1 % over the first 16 iterations, 2 % at
    any of the 40 (measured: 0.29 % max) (tolerance on the loss, not on bits: two fp32 runs with different reduction orders drift apart too)."""
    import json
    import os
    from conftest import ROOT
    steps, bs, size = 40, 32, 640              # the config's own per-GPU batch (WIDERFACE_LFD_S.py: batch 32 per GPU)

    def batch(it):                              # regenerated per iteration from its own seed: 40 x 157 MB would not fit the host
        rng = np.random.default_rng(5500 + it)
        x = torch.randn((bs, 3, size, size), device='cuda', generator=torch.Generator(device='cuda').manual_seed(5500 + it))
        ann = []
        for _ in range(bs):
            g = int(rng.integers(1, 21))
            wh = np.exp(rng.uniform(np.log(6), np.log(300), (g, 2)))
            xy = rng.uniform(0, 1, (g, 2)) * (size - wh).clip(1)
            ann.append((np.concatenate([xy, wh], 1).astype(np.float32), np.zeros(g, np.int64)))
        return x, ann
    clip = dict(max_norm=10, norm_type=2)
    curves, norms = {}, {}
    for mode in ('hip', 'torch'):
        monkeypatch.setenv('LFD_HIP_TRAIN', '1' if mode == 'hip' else '0')
        monkeypatch.setenv('LFD_FUSED_LOSS', '1' if mode == 'hip' else '0')
        m = configs.build_model('WIDERFACE_LFD_S', seed=666).cuda().train()
        kw = dict(lr=0.1, momentum=0.9, weight_decay=1e-4)
        opt = optim.SGD(m.parameters(), **kw) if mode == 'hip' else torch.optim.SGD(m.parameters(), **kw)
        losses, gns = [], []
        scaler = train.DynamicLossScale(init_scale=1024.0) if mode == 'hip' else None
        if scaler is not None:
            hip_scaler = scaler
        for it in range(steps):
            x, ann = batch(it)
            lr = 0.1 * (0.1 + (1 - 0.1) * it / 200.0)          # LrSchedulerHook linear warm-up (lr_scheduler_hook.py:80-99)
            for gr in opt.param_groups:
                gr['lr'] = lr
            lv, gn = train.train_step(m, opt, x.cuda(), ann, clip, clip_active=True, loss_scaler=scaler)
            assert np.isfinite(lv['loss']) and np.isfinite(float(gn)), (mode, it)
            losses.append(float(lv['loss']))
            gns.append(float(gn))
        curves[mode], norms[mode] = losses, gns
    h, t = np.array(curves['hip']), np.array(curves['torch'])
    rel = np.abs(h - t) / np.abs(t)
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    json.dump(dict(hip=curves['hip'], torch=curves['torch'], grad_norm_hip=norms['hip'], grad_norm_torch=norms['torch'],
                   rel=rel.tolist()), open(os.path.join(out, 'train_curve_config5.json'), 'w'), indent=1)
    print('config 5 loss curve: hip %s\n torch %s\n max rel diff %.3e (first step %.3e)' % (np.round(h, 4).tolist(), np.round(t, 4).tolist(),
                                                                                        rel.max(), rel[0]))
    assert hip_scaler.skipped == 0 and hip_scaler.scale == 1024.0  # no overflow at the default loss scale
    assert rel[0] < 5e-3                                  # same start: only the fp16 forward differs
    assert rel[:16].max() < 1e-2 and rel.max() < 2e-2     # asked: 2 % / 5 % at bs 32 over 40 iterations; measured 0.29 % max
    # the total gradient norm the two routes report: within 2 % over the first 15 iterations (measured
    # <= 1.2 %: 413.779 / 413.779, 151.020 / 151.025, 11.874 / 11.880, ...); later the two TRAJECTORIES have separated enough
    # that a norm of ~1 is compared between different weights (ratio 0.86 .. 1.23 after iteration 25) -- what kernel error
    # contributes to that, iteration by iteration from the same state, is gated in tests/test_train_golden.py (<= 0.81 %)
    # (the fp32 comparator is not bit-reproducible from run to run -- MIOpen's backward kernels -- so iterations 11-15, where
    #  the norms have fallen from 414 to ~3 and two trajectories start to differ, get 4 %: one of three runs of round 4 exceeded
    #  2 % there, the other two measured 1.1 % / 1.2 %)
    gh, gt = np.array(norms['hip']), np.array(norms['torch'])
    dev = np.abs(gh[:15] - gt[:15]) / gt[:15]
    assert dev[:10].max() < 2e-2 and dev.max() < 4e-2, (gh[:15], gt[:15])
    assert np.mean(h[-4:]) < h[0]                         # and go down


@pytest.mark.parametrize('bad', [float('inf'), float('-inf'), float('nan')])
def test_update_is_skipped_when_the_gradient_norm_is_not_finite(bad):
    """ADVICE r2 (csrc/optim.hip): an fp16 overflow in the activation-gradient path yields +-inf gradients WITHOUT a NaN:
    norm = inf, clip coefficient = max_norm / inf = 0 -- inside [0, 1] -- and g * coef = inf * 0 = NaN would be written
    into the weights and the momentum.  The update kernel gates on the NORM: parameters, momentum buffers and gradients
    stay exactly as they were, with clipping and without; the next clean iteration updates normally; DynamicLossScale
    halves the scale on the skipped iteration."""
    from lfd_amd import train_engine
    ma, mb = _pair()
    oa = optim.SGD(ma.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    ob = torch.optim.SGD(mb.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    oa.zero_grad()
    _set_grads(ma, mb, 5, 1.0)
    oa.clip_and_step(10.0)                      # a clean first step: momentum buffers exist from here on
    torch.nn.utils.clip_grad_norm_(list(mb.parameters()), max_norm=10.0, norm_type=2)
    ob.step()
    scaler = train.DynamicLossScale(init_scale=1024.0, growth_interval=2)
    try:
        for clip in (True, False):
            _set_grads(ma, mb, 6, 1.0)
            p0 = list(ma.parameters())[3]
            p0.grad.view(-1)[7] = bad
            before = [(p.detach().clone(), oa.state[p]['momentum_buffer'].clone(), p.grad.clone()) for p in ma.parameters()]
            norm = oa.clip_and_step(10.0) if clip else (oa.step(), oa.last_norm[0])[1]
            assert not np.isfinite(float(norm))
            for p, (pv, mv, gv) in zip(ma.parameters(), before):
                assert torch.equal(p.detach(), pv) and torch.equal(oa.state[p]['momentum_buffer'], mv)
                assert torch.equal(p.grad, gv, ) or (bad != bad and bool(torch.isnan(p.grad).any()))
                assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(oa.state[p]['momentum_buffer']).all())
            s0 = scaler.scale
            assert scaler.update(norm) is False and scaler.scale == s0 / 2
        # clean iterations: identical to torch again (the skipped ones left no trace), and the scale grows back
        for it in range(2):
            _set_grads(ma, mb, 8 + it, 0.5)
            na = oa.clip_and_step(10.0)
            torch.nn.utils.clip_grad_norm_(list(mb.parameters()), max_norm=10.0, norm_type=2)
            ob.step()
            assert scaler.update(na) is True
            for pa, pb in zip(ma.parameters(), mb.parameters()):
                torch.testing.assert_close(pa, pb, rtol=3e-6, atol=2e-8)
        assert scaler.scale == 512.0 and scaler.skipped == 2
    finally:
        train_engine.set_loss_scale(train_engine.LOSS_SCALE)


def test_graphed_step_survives_a_loss_scale_cycle_and_a_second_batch_shape():
    """ADVICE r4 (high / medium): a captured training graph holds the raw addresses of the batched-finals job table and of the
    persistent partial buffers.  Scale A -> B -> A (what a DynamicLossScale does after an overflow) and an eager step of
    another batch shape in between must leave the first graph's table and buffers alive: the replay at scale A afterwards
    equals an uninterrupted run."""
    from lfd_amd import configs, optim, train, train_engine
    name = 'WIDERFACE_LFD_XS'
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(2, 3, 96, 128, device='cuda', generator=g)
    x_other = torch.randn(1, 3, 64, 96, device='cuda', generator=g)
    ann = [(np.array([[10., 12., 30., 40.], [60., 20., 50., 44.]], np.float32), np.zeros(2, np.int64)),
           (np.array([[40., 30., 24., 20.]], np.float32), np.zeros(1, np.int64))]
    clip = dict(max_norm=10, norm_type=2)
    keep = train_engine.loss_scale()

    def run(disturb):
        torch.manual_seed(666)
        m = configs.build_model(name).cuda().train()
        opt = optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        train_engine.set_loss_scale(1024.0)
        step = train.GraphedTrainStep(m, opt, clip, max_boxes=64)
        out = []
        for it in range(6):
            if disturb and it == 3:
                # an "overflow": one iteration at half the scale (eager: new key), one eager step of another batch shape
                # on a scratch copy of the model's plan owner, then back to the captured scale
                train_engine.set_loss_scale(512.0)
                step(x, ann, True)
                train_engine.set_loss_scale(1024.0)
            lv, gn = step(x, ann, True)
            out.append((lv['loss'], float(gn)))
        return out, torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    try:
        a, pa = run(False)
        # the disturbed run does one extra iteration (at scale 512: same arithmetic up to fp16 gradient rounding), so compare
        # the runs structurally: the replay after the cycle must be finite, close to the undisturbed trajectory and
        # the job tables / buffers of the first graph must not have been rebuilt in place
        b, pb = run(True)
        assert all(np.isfinite(v) for t in b for v in t)
        # (the disturbed run made one extra update at it == 3: its entry k >= 3 is the undisturbed run's entry k + 1)
        assert abs(b[-2][0] - a[-1][0]) <= 0.02 * abs(a[-1][0]), (a, b)
        # direct check of the mechanism: tables are kept per content, buffers per (key, size)
        m = configs.build_model(name).cuda().train()
        opt = optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        train.train_step(m, opt, x, ann, clip, True)
        sc = m.__dict__['_lfd_train_sched']
        tabs = {k: v[0].data_ptr() for k, v in sc.finals.cache.items()}
        bufs = {k: v.data_ptr() for k, v in sc.bufs.items()}
        train_engine.set_loss_scale(256.0)
        train.train_step(m, opt, x, ann, clip, True)
        train.train_step(m, opt, x_other, ann[:1], clip, True)
        train_engine.set_loss_scale(1024.0)
        train.train_step(m, opt, x, ann, clip, True)
        for k, pnt in tabs.items():
            assert sc.finals.cache[k][0].data_ptr() == pnt
        for k, pnt in bufs.items():
            assert sc.bufs[k].data_ptr() == pnt
        assert len(sc.finals.cache) > len(tabs) and len(sc.bufs) > len(bufs)
    finally:
        train_engine.set_loss_scale(keep)
