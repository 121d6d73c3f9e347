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


# ---------------------------------------------------------------------------------------------- GPU tier
# Gates of the free-running comparison, (loss rtol, gradient-norm rtol) per iteration + running-statistics rtol after the third.
# Iteration 1 starts from the reference's own weights: 1 % / 3 % for every case (measured <= 0.08 % / 0.5 %).  From iteration 2
# on the comparison is between two TRAJECTORIES, and these iterations are a violent transient by construction (perturbed
# weights, loss 14.0 -> 5.0 -> 1.8, gradient norm 163 -> 59 -> 19 for WIDERFACE_LFD_S): on the MI355X the FP32 route of
# this package (PyTorch-ROCm autograd over the mirror modules, equal to the reference to 1e-4 in iteration 1) is itself 0.6 %
# off the reference's gradient norm in iteration 3 -- a x 60 amplification of rounding-level differences in two updates.
# The HIP route enters with fp16-storage differences (0.5 % on the first gradient norm) and leaves the tiny cases at 13 %
# (measured, tools/train_golden_diag.py -> gpurun_out/train_golden.json, round 4; VERDICT r3 item 1: this -- not a kernel
# defect -- is what failed behind the old xfail: WIDERFACE_LFD_S gradient norm 13.6 % against a 10 % estimate, WIDERFACE_LFD_XS
# classification loss 3.4 % against 3 %).  The tiny cases keep measured gates (value x ~1.5); the LARGE cases -- every
# BatchNorm sees >= 512 elements per channel -- carry the 1 % / 3 % gates, except the third gradient norm of the 32-channel-stem
# model (gradient norm 1344 -> 335 -> 21.7: measured 3.5 %, gate 6 %).  What separates kernel error from trajectory
# divergence is the teacher-forced test below: every iteration started from the FP32 route's state, gated at 1 % / 3 %.
_FREE_GATES = {
    'WIDERFACE_LFD_S': ([0.01, 0.015, 0.03], [0.03, 0.03, 0.20], 0.01),           # measured 0.0004 0.0073 0.0188 | 0.005 0.015 0.136 | 0.0055
    'WIDERFACE_LFD_XS': ([0.01, 0.01, 0.05], [0.03, 0.03, 0.20], 0.03),           # 0.0008 0.0040 0.0339 | 0.0002 0.008 0.129 | 0.019
    'TT100K_LFD_L': ([0.01, 0.01, 0.01], [0.03, 0.03, 0.03], 0.01),               # 0.0001 0.0004 0.0031 | 0.0006 0.0004 0.0059 | 0.0038
    'TL_LFD_L': ([0.01, 0.01, 0.01], [0.03, 0.03, 0.03], 0.03),                   # 0.0000 0.0000 0.0001 | 0.0001 0.0001 0.0001 | 0.0026-0.0103 (neck4: 2 x 1 x 2 values)
    'WIDERFACE_LFD_S@8x512x512': ([0.01, 0.01, 0.01], [0.03, 0.03, 0.03], 0.003),   # 0.0005 0.0013 0.0040 | 0.0001 0.0002 0.0039 | 0.00016
    'WIDERFACE_LFD_XS@8x512x512': ([0.01, 0.01, 0.01], [0.03, 0.03, 0.06], 0.003),  # 0.0005 0.0003 0.0014 | 0.0003 0.0003 0.0353 | 0.00015
}


def _record(key, value):
    """measured values of this file's GPU tests -> gpurun_out/train_golden_tests.json (merged across tests)"""
    import json
    import os
    from conftest import ROOT
    d = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(d, exist_ok=True)
    f = os.path.join(d, 'train_golden_tests.json')
    try:
        cur = json.load(open(f))
    except Exception:
        cur = {}
    cur[key] = value
    json.dump(cur, open(f, 'w'), indent=1)


def _golden(name):
    return load_golden('ref_train_step_%s.npz' % cases.file_tag(name))


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(cases.CASES) + list(cases.LARGE_CASES))
def test_hip_training_path_follows_the_reference_iterations(name):
    """train.train_step on the HIP path (train_engine forward / backward, fused get_loss, flat SGD with fused clipping), three
    free-running iterations against the REAL reference's: losses, gradient norms, BatchNorm running statistics after the
    third.  fp16 activations against the reference's fp32; gates and what they mean: _FREE_GATES above."""
    from lfd_amd import optim
    g = _golden(name)
    arch = cases.shape_of(name)[0]
    m = _model(arch).cuda()
    opt = optim.SGD(m.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    clip = {k: v for k, v in cases.GRAD_CLIP.items() if k != 'duration'}
    x = cases.images(name).cuda()
    ann = cases.annotations(name, configs.ARCHS[arch]['num_classes'])
    loss_gate, norm_gate, stat_gate = _FREE_GATES[name]
    rec = dict(loss_rel_err=[], grad_norm_rel_err=[])
    fails = []
    for it in range(cases.ITERATIONS):
        lv, gn = train.train_step(m, opt, x, ann, clip, clip_active=True)
        want = g['losses'][it]
        got = np.array([float(lv['loss']), float(lv['classification_loss']), float(lv['regression_loss'])])
        le = np.abs(got - want) / np.abs(want)
        ne = abs(float(gn) - float(g['grad_norms'][it])) / float(g['grad_norms'][it])
        rec['loss_rel_err'].append(le.tolist())
        rec['grad_norm_rel_err'].append(ne)
        if le.max() > loss_gate[it]:
            fails.append('iteration %d losses %s vs %s' % (it + 1, got.tolist(), want.tolist()))
        if ne > norm_gate[it]:
            fails.append('iteration %d gradient norm %g vs %g' % (it + 1, float(gn), float(g['grad_norms'][it])))
    sd = m.state_dict()
    worst = 0.0
    for k, row, w in zip(sd.keys(), [_summary(v) for v in sd.values()], g['state_summary_%d' % (cases.ITERATIONS - 1)]):
        if k.endswith('running_mean') or k.endswith('running_var'):
            e = abs(row[0] - w[0]) / max(w[0], 1e-3)
            worst = max(worst, e)
            if e > stat_gate:
                fails.append('%s %g vs %g' % (k, row[0], w[0]))
        if k.endswith('num_batches_tracked'):
            assert row[2] == w[2] == cases.ITERATIONS, k
    rec['running_stat_rel_err_worst'] = worst
    _record('free_running/' + name, rec)
    assert not fails, fails


def _sync_state(ma, oa, mb, ob):
    """model / optimizer B <- A: parameters, BatchNorm buffers, momentum buffers (in place: B's stay views of its flat buffers)"""
    with torch.no_grad():
        for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert ka == kb
            vb.copy_(va)
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            ba = oa.state.get(pa, {}).get('momentum_buffer')
            if ba is not None:
                ob.state[pb]['momentum_buffer'].copy_(ba)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['WIDERFACE_LFD_S', 'WIDERFACE_LFD_XS', 'TT100K_LFD_L', 'WIDERFACE_LFD_S@8x512x512',
                                  'WIDERFACE_LFD_XS@8x512x512'])
def test_every_hip_iteration_from_the_fp32_routes_state(name, monkeypatch):
    """Teacher-forced form of the test above -- kernel error without trajectory divergence.  Route A = the mirror modules
    through PyTorch-ROCm fp32 autograd + torch.optim.SGD (LFD_HIP_TRAIN=0), which the CPU suite pins to the reference bit for
    bit; route B = the HIP training path.  Before EVERY iteration B receives A's parameters, BatchNorm buffers and momentum
    buffers, then both run the iteration.  Gates: A against the reference's iterations 1 % loss / 3 % gradient norm (measured
    <= 0.3 % / 0.6 %, the fp32 trajectory's own drift); B against A, same state: **1 % loss, 3 % gradient norm** (measured on
    the MI355X, round 4: <= 0.12 % / <= 0.81 % over every case and iteration), BatchNorm running statistics after the
    iteration 1 % (tiny cases, measured 0.55 %) / 0.1 % (large, 0.04 %), and the cosine between the two whole gradient vectors:
    >= 0.98 for the tiny cases (measured 0.987-0.998: ~0.5 % of the ReLU decisions of an fp16-storage forward differ from
    the fp32 forward's, each a 100 % error of that element's gradient -- DESIGN 4), >= 0.999 for the large cases in iterations
    1-2 (measured 0.9998-0.9999) and >= 0.96 in iteration 3 (0.972 / 0.982: by then the gradient is the small residual of a
    nearly converged classification loss -- norm 1783 -> 44 -- and the same flipped elements weigh 40 x more)."""
    from lfd_amd import optim
    g = _golden(name)
    arch = cases.shape_of(name)[0]
    ma, mb = _model(arch).cuda(), _model(arch).cuda()
    oa = torch.optim.SGD(ma.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    ob = optim.SGD(mb.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    x = cases.images(name).cuda()
    ann = cases.annotations(name, configs.ARCHS[arch]['num_classes'])
    max_norm = float(cases.GRAD_CLIP['max_norm'])
    rec = dict(a_vs_ref_loss=[], a_vs_ref_norm=[], b_vs_a_loss=[], b_vs_a_norm=[], b_vs_a_cos=[], b_vs_a_stats=[])
    fails = []
    for it in range(cases.ITERATIONS):
        _sync_state(ma, oa, mb, ob)
        monkeypatch.setenv('LFD_HIP_TRAIN', '0')
        la = ma.get_loss(ma(x), ann)
        oa.zero_grad()
        la['loss'].backward()
        ga = torch.cat([p.grad.reshape(-1) for p in ma.parameters()]).double()
        na = float(torch.nn.utils.clip_grad_norm_(list(ma.parameters()), max_norm, 2))
        oa.step()
        monkeypatch.setenv('LFD_HIP_TRAIN', '1')
        lb = mb.get_loss(mb(x), ann)
        ob.zero_grad()
        lb['loss'].backward()
        gb = torch.cat([p.grad.reshape(-1) for p in mb.parameters()]).double()
        nb = float(ob.clip_and_step(max_norm))
        va = np.array([la['loss_values'][k] for k in ('loss', 'classification_loss', 'regression_loss')], np.float64)
        vb = np.array([lb['loss_values'][k] for k in ('loss', 'classification_loss', 'regression_loss')], np.float64)
        e_ref = float((np.abs(va - g['losses'][it]) / np.abs(g['losses'][it])).max())
        n_ref = abs(na - float(g['grad_norms'][it])) / float(g['grad_norms'][it])
        e_ab = float((np.abs(vb - va) / np.abs(va)).max())
        n_ab = abs(nb - na) / na
        cos = float(ga @ gb / (ga.norm() * gb.norm()))
        st = 0.0
        for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
            if k.endswith('running_mean') or k.endswith('running_var'):
                st = max(st, float((a - b).double().norm() / a.double().norm().clamp_min(1e-3)))
        for key, v in (('a_vs_ref_loss', e_ref), ('a_vs_ref_norm', n_ref), ('b_vs_a_loss', e_ab), ('b_vs_a_norm', n_ab),
                       ('b_vs_a_cos', cos), ('b_vs_a_stats', st)):
            rec[key].append(v)
        large = name in cases.LARGE_CASES
        cos_gate = (0.04 if it == 2 else 0.001) if large else 0.02
        for what, v, gate in (('fp32 route vs reference, loss', e_ref, 0.01), ('fp32 route vs reference, gradient norm', n_ref, 0.03),
                              ('HIP vs fp32 route, loss', e_ab, 0.01), ('HIP vs fp32 route, gradient norm', n_ab, 0.03),
                              ('HIP vs fp32 route, 1 - cosine of the gradient', 1 - cos, cos_gate),
                              ('HIP vs fp32 route, running statistics', st, 0.001 if large else 0.01)):
            if v > gate:
                fails.append('iteration %d: %s %.4g > %g' % (it + 1, what, v, gate))
    _record('teacher_forced/' + name, rec)
    assert not fails, fails
