"""tests/golden/make_golden_train_step.py -- three TRAINING ITERATIONS of the real reference, frozen as fixtures.

    python tests/golden/make_golden_train_step.py [case ...]

Runs, on CPU in the build container, the loop body of the reference's Executor.train (lfd/execution/executor.py:191-211:
model(image_batch) in train mode -> model.get_loss -> hooks) with the reference's OWN OptimizerHook.after_train_iter
(lfd/execution/hooks/optimizer_hook.py:26-36: zero_grad, loss.backward(), clip_grad_norm_ during the first `duration`
epochs, optimizer.step()) and the optimizer the config files build (WIDERFACE_LFD_S.py:217-226: torch.optim.SGD,
momentum 0.9, weight decay 1e-4, grad clip max_norm 10 / norm_type 2 / duration 5; lr = the config's 0.1 x its warm-up
ratio 0.1, the first iteration's value), from seeded weights, on one seeded image batch with seeded annotations.

Output (committed): ref_train_step_<ARCH>.npz with, per iteration: the three loss values, the total gradient norm the hook
reports; from iteration 1: the train-mode outputs (cls, reg), dL/dcls and dL/dreg, per-parameter gradient summaries
(L2 norm, mean, first four elements) and the full gradients of every 1-D parameter; after iteration 1 and 3: per-tensor
summaries of the whole state_dict (parameters after the update, BatchNorm running statistics, num_batches_tracked).
Weights are not stored: torch.manual_seed(666) + configs.perturb_weights(seed=1) rebuild them (the sha256 of the
reference state_dict is stored, as in make_golden.py).  Consumers: tests/test_train_golden.py.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
sys.path.insert(0, HERE)
warnings.filterwarnings('ignore')

from oracle import ref_import  # noqa: E402
from lfd_amd import configs  # noqa: E402  (only the arch dicts + perturbation helper)
from make_golden import state_sha  # noqa: E402
import train_step_cases as cases  # noqa: E402  (seeded inputs shared with the tests)


def summary(t):
    """[L2 norm, mean, first four elements] of a tensor, float64"""
    f = t.detach().double().reshape(-1)
    head = torch.zeros(4, dtype=torch.float64)
    head[:min(4, f.numel())] = f[:4]
    return np.concatenate([[float(f.norm()), float(f.mean())], head.numpy()])


def main():
    M = ref_import.import_reference()
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN
    # the reference's hook module itself, without its package __init__s (lfd/execution/__init__.py pulls the executor, the
    # logger hook and with them torchvision, which this image does not have): parents pre-seeded as bare packages
    import importlib
    import types
    for pkg, sub in (('lfd.execution', 'lfd/execution'), ('lfd.execution.hooks', 'lfd/execution/hooks')):
        if pkg not in sys.modules:
            mod = types.ModuleType(pkg)
            mod.__path__ = [os.path.join(ref_import.REF_ROOT, sub)]
            sys.modules[pkg] = mod
    RefHook = importlib.import_module('lfd.execution.hooks.optimizer_hook').OptimizerHook

    only = sys.argv[1:]
    for name in list(cases.CASES) + list(cases.LARGE_CASES):
        if only and name not in only:
            continue
        large = name in cases.LARGE_CASES                              # (round 4) summaries only, see train_step_cases.py
        arch = configs.ARCHS[cases.shape_of(name)[0]]
        model = configs.build_modules(arch, RB.LFDResNet, RN.SimpleNeck, RH.LFDHead, M.LFD, RL.FocalLoss,
                                      RL.IoULoss, RL.CrossEntropyLoss, seed=666, qfl_cls=RL.QualityFocalLoss)
        configs.perturb_weights(model, seed=1)
        res = dict(sha=state_sha(model.state_dict()))
        model.train()
        x = cases.images(name)
        ann = cases.annotations(name, arch['num_classes'])
        opt = torch.optim.SGD(model.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
        hook = RefHook(dict(cases.GRAD_CLIP), training_epochs=1000)

        class Executor(object):
            config_dict = dict(model=model, optimizer=opt, epoch=0)

        names = [k for k, _ in model.named_parameters()]
        losses, norms = [], []
        for it in range(cases.ITERATIONS):
            cls, reg = model(x)                                         # executor.py:201
            if it == 0:
                cls.retain_grad()
                reg.retain_grad()
                if large:      # every 8th point of the train-mode outputs
                    res['cls_s8'], res['reg_s8'] = cls.detach().numpy()[:, ::8].copy(), reg.detach().numpy()[:, ::8].copy()
                else:
                    res['cls'], res['reg'] = cls.detach().numpy().copy(), reg.detach().numpy().copy()
            lo = model.get_loss((cls, reg), ann)                         # executor.py:203-205
            Executor.config_dict.update(loss=lo['loss'])
            hook.after_train_iter(Executor)                              # optimizer_hook.py:26-36
            lv = lo['loss_values']
            losses.append([lv['loss'], lv['classification_loss'], lv['regression_loss']])
            norms.append(float(Executor.config_dict['grad_norm']))
            if it == 0:
                # (the hook clipped p.grad in place: undo the clip coefficient so that the stored gradients are dL/dp)
                coef = min(1.0, cases.GRAD_CLIP['max_norm'] / (norms[0] + 1e-6))
                if not large:
                    res['dcls'], res['dreg'] = cls.grad.numpy().copy(), reg.grad.numpy().copy()
                res['grad_summary'] = np.stack([summary(p.grad / coef) for _, p in model.named_parameters()])
                for k, p in model.named_parameters():
                    if p.dim() <= 1:
                        res['grad/' + k] = (p.grad / coef).detach().numpy().copy()
            if it in (0, cases.ITERATIONS - 1):
                sd = model.state_dict()
                res['state_summary_%d' % it] = np.stack([summary(v) for v in sd.values()])
        res['param_names'] = np.array(names)
        res['state_names'] = np.array(list(model.state_dict().keys()))
        res['losses'] = np.array(losses, np.float64)
        res['grad_norms'] = np.array(norms, np.float64)
        res['sizes'] = np.array([model.head_indexes_to_feature_map_sizes[i] for i in range(len(arch['regression_ranges']))])
        np.savez_compressed(os.path.join(os.environ.get('LFD_GOLDEN_OUT', HERE), 'ref_train_step_%s.npz' % cases.file_tag(name)), **res)
        print(name, 'losses', res['losses'][:, 0], 'grad norms', res['grad_norms'], 'P', cls.shape[1])


if __name__ == '__main__':
    main()
