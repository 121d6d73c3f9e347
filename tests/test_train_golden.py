"""Training iterations of the REAL reference (tests/golden/make_golden_train_step.py: Executor.train's loop body,
executor.py:191-211, with the reference's own OptimizerHook, optimizer_hook.py:26-36, and the SGD its configs build) against
this repository, on the same seeded weights, images and annotations (tests/golden/train_step_cases.py).

CPU tier -- what runs without a device: the train-mode forward of the module mirrors (BatchNorm batch statistics, GroupNorm
towers, Scale), their autograd given the reference's dL/dcls and dL/dreg, this repository's OptimizerHook with the same
optimizer: outputs, every parameter gradient, the reported gradient norm and the whole state_dict after the update
(BatchNorm running statistics and num_batches_tracked included) equal the reference's -- outputs bit for bit, gradients to
3e-7, the updated state to 4e-9 in the build container.  (The loss itself has no CPU path here by design -- its kernels
are pinned to reference goldens in tests/test_gpu_losses.py.)  This is what makes
"the mirror modules through PyTorch autograd" -- the comparator of the GPU training tests -- a stand-in for the reference.

GPU tier -- the HIP training path (train_engine + fused loss + flat SGD) run for the same three iterations."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from lfd_amd import configs, train
import train_step_cases as cases


def _sha(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def _summary(t):
    f = t.detach().double().reshape(-1).cpu()
    head = torch.zeros(4, dtype=torch.float64)
    head[:min(4, f.numel())] = f[:4]
    return np.concatenate([[float(f.norm()), float(f.mean())], head.numpy()])


def _model(name):
    m = configs.build_model(name)                 # torch.manual_seed(666), the config seed
    configs.perturb_weights(m, seed=1)
    return m.train()


def _close_summaries(got, want, names, rtol, what):
    """rows [L2 norm, mean, first four elements]: element-wise, absolute tolerance relative to the tensor's RMS-ish scale"""
    for g, w, k in zip(got, want, names):
        scale = max(abs(w[0]), 1e-12)
        assert abs(g[0] - w[0]) <= rtol * scale, (what, k, 'norm', g[0], w[0])
        np.testing.assert_allclose(g[1:], w[1:], rtol=rtol, atol=rtol * scale, err_msg='%s %s' % (what, k))


@pytest.mark.parametrize('name', list(cases.CASES))
def test_mirror_modules_and_hook_reproduce_the_reference_training_iteration_on_cpu(name):
    g = load_golden('ref_train_step_%s.npz' % name)
    m = _model(name)
    assert _sha(m.state_dict()) == str(g['sha'])                       # same initial weights as the reference model had
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in g['param_names']]
    cls, reg = m(cases.images(name))                                      # CPU tensors: the PyTorch module path
    assert [tuple(m.head_indexes_to_feature_map_sizes[i]) for i in range(len(g['sizes']))] == [tuple(s) for s in g['sizes'].tolist()]
    # (measured in the build container: bit-identical; the gate leaves room for another host's thread count in the CPU convs)
    np.testing.assert_allclose(cls.detach().numpy(), g['cls'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(reg.detach().numpy(), g['reg'], rtol=1e-5, atol=2e-6)
    # the iteration's backward + update through this repository's hook: a loss that is linear in the outputs with the
    # reference's dL/dcls, dL/dreg as coefficients has exactly the reference loss's gradients
    loss = (cls * torch.from_numpy(g['dcls'])).sum() + (reg * torch.from_numpy(g['dreg'])).sum()
    opt = torch.optim.SGD(m.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    hook = train.OptimizerHook(dict(cases.GRAD_CLIP), training_epochs=1000)

    class Executor(object):
        config_dict = dict(model=m, optimizer=opt, loss=loss, epoch=0)
    hook.after_train_iter(Executor)
    norm = float(Executor.config_dict['grad_norm'])
    assert norm == pytest.approx(float(g['grad_norms'][0]), rel=1e-5)                      # (measured: equal)
    coef = min(1.0, cases.GRAD_CLIP['max_norm'] / (norm + 1e-6))          # p.grad was clipped in place
    names = [k for k, _ in m.named_parameters()]
    _close_summaries([_summary(p.grad / coef) for _, p in m.named_parameters()], g['grad_summary'], names, 2e-5, 'gradient')   # (2.7e-7)
    for k, p in m.named_parameters():
        if p.dim() <= 1:
            w = g['grad/' + k]
            np.testing.assert_allclose((p.grad / coef).numpy(), w, rtol=2e-5, atol=2e-5 * float(np.abs(w).max() + 1e-12), err_msg=k)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['state_names']]
    _close_summaries([_summary(v) for v in sd.values()], g['state_summary_0'], list(sd.keys()), 1e-6, 'state after the update')   # (4e-9)


@pytest.mark.parametrize('name', list(cases.CASES))
def test_loss_oracle_on_the_training_iteration_vectors(name):
    """oracle/net_oracle.lfd_loss (the checker of the fused loss kernels and of smoke()'s training step) on a second set of
    reference vectors: the train-mode outputs and annotations of iteration 1 -> the reference's three loss values."""
    from oracle import net_oracle
    g = load_golden('ref_train_step_%s.npz' % name)
    arch = configs.ARCHS[name]
    ann = cases.annotations(name, arch['num_classes'])
    out = net_oracle.lfd_loss(arch, torch.from_numpy(g['cls']), torch.from_numpy(g['reg']), [tuple(s) for s in g['sizes'].tolist()],
                              net_oracle.strides_of(arch), [a[0] for a in ann], [a[1] for a in ann])
    np.testing.assert_allclose([out['loss'], out['classification_loss'], out['regression_loss']], g['losses'][0], rtol=3e-6)


@pytest.mark.gpu
@pytest.mark.xfail(strict=False, reason='written after round 3\'s GPU minutes were spent: gates are estimates from the other '
                                        'training tests (loss curve within 3 %), first hardware run pending')
@pytest.mark.parametrize('name', list(cases.CASES))
def test_hip_training_path_follows_the_reference_iterations(name):
    """train.train_step on the HIP path (train_engine forward / backward, fused get_loss, flat SGD with fused clipping), three
    iterations: losses and gradient norms against the reference's, BatchNorm running statistics after the third.  fp16
    activations against the reference's fp32: the gates are those of the config-5 loss-curve test, not rounding."""
    from lfd_amd import optim
    g = load_golden('ref_train_step_%s.npz' % name)
    m = _model(name).cuda()
    opt = optim.SGD(m.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    clip = {k: v for k, v in cases.GRAD_CLIP.items() if k != 'duration'}
    x = cases.images(name).cuda()
    ann = cases.annotations(name, configs.ARCHS[name]['num_classes'])
    for it in range(cases.ITERATIONS):
        lv, gn = train.train_step(m, opt, x, ann, clip, clip_active=True)
        want = g['losses'][it]
        got = [float(lv['loss']), float(lv['classification_loss']), float(lv['regression_loss'])]
        np.testing.assert_allclose(got, want, rtol=3e-2, err_msg='iteration %d losses' % it)
        assert float(gn) == pytest.approx(float(g['grad_norms'][it]), rel=0.1), it
    sd = m.state_dict()
    for k, row, w in zip(sd.keys(), [_summary(v) for v in sd.values()], g['state_summary_%d' % (cases.ITERATIONS - 1)]):
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert abs(row[0] - w[0]) <= 3e-2 * max(w[0], 1e-3), k
        if k.endswith('num_batches_tracked'):
            assert row[2] == w[2] == cases.ITERATIONS, k
