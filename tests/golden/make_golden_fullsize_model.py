"""tests/golden/make_golden_fullsize_model.py -- the REAL reference at the BASELINE shapes themselves.

    python tests/golden/make_golden_fullsize_model.py

(1) LFD.forward (lfd.py:511-542) of configs 2 / 3 / 4 on one seeded full-size frame (1920 x 1080, 3840 x 2160, 1280 x 720),
seeded + perturbed weights as in make_golden.py: the per-level sizes it records, a strided sample (every 37th point) of the
cls / reg tensors and their float64 sums -- the small fixtures run the same code on 72 x 104 ... 96 x 128 frames, where no
level is wider than 32 points and the 64 -> 128 stage sees 2 x 3 maps.
(2) config 5's target assignment and loss at its own size (lfd.py:109-259, :284-395): WIDERFACE_LFD_S, 32 images of
640 x 640, 1-20 boxes each (fullsize_cases.train_annotations), 8,600 points per image: sha256 of the classification /
regression target tensors, positive / gray counts, and get_loss's three values + prediction-gradient norms on seeded logits.
Output: ref_fullsize_model.npz.  Consumer: tests/test_oracle_golden.py."""
import hashlib
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
import fullsize_cases as cases  # noqa: E402


def build(M, name):
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN
    model = configs.build_modules(configs.ARCHS[name], RB.LFDResNet, RN.SimpleNeck, RH.LFDHead, M.LFD, RL.FocalLoss,
                                  RL.IoULoss, RL.CrossEntropyLoss, seed=666, qfl_cls=RL.QualityFocalLoss)
    configs.perturb_weights(model, seed=1)
    return model


def main():
    M = ref_import.import_reference()
    out = {}
    for key, (name, (H, W), sizes) in cases.GRIDS.items():
        model = build(M, name).eval()
        with torch.no_grad():
            cls, reg = model(cases.frame(key))
        got = [tuple(model.head_indexes_to_feature_map_sizes[i]) for i in range(len(sizes))]
        assert got == [tuple(s) for s in sizes], (key, got)           # fullsize_cases.GRIDS states what the forward records
        idx = cases.sample_index(cls.shape[1])
        out[key + '/cls'], out[key + '/reg'] = cls[0, idx].numpy(), reg[0, idx].numpy()
        out[key + '/sums'] = np.array([float(cls.double().sum()), float(reg.double().sum()),
                                       float(cls.double().abs().sum()), float(reg.double().abs().sum())])
        print(key, name, tuple(cls.shape), 'sample', len(idx), 'sums', out[key + '/sums'])
    # ---- config 5
    name, n, h, w, sizes = cases.TRAIN
    model = build(M, name).eval()
    for i, s in enumerate(sizes):
        model._head_indexes_to_feature_map_sizes[i] = tuple(s)
    ann = cases.train_annotations()
    pts = model.generate_point_coordinates(model.head_indexes_to_feature_map_sizes)
    ct, rt = model.annotation_to_target(pts, [torch.from_numpy(a[0]) for a in ann], [torch.from_numpy(a[1]) for a in ann])
    assert tuple(ct.shape) == (n, sum(a * b for a, b in sizes), 1)
    out['train/cls_target_sha'] = hashlib.sha256(ct.numpy().tobytes()).hexdigest()
    out['train/reg_target_sha'] = hashlib.sha256(rt.numpy().tobytes()).hexdigest()
    out['train/counts'] = np.array([int((ct > 0).sum()), int((ct < 0).sum()), int((ct >= 0.001).sum()), sum(len(a[1]) for a in ann)])
    out['train/target_sums'] = np.array([float(ct.double().sum()), float(rt.double().sum())])
    cl, rg = cases.train_logits()
    cg, rgg = torch.from_numpy(cl).requires_grad_(True), torch.from_numpy(rg).requires_grad_(True)
    lo = model.get_loss((cg, rgg), ann)
    lo['loss'].backward()
    lv = lo['loss_values']
    out['train/loss'] = np.array([lv['loss'], lv['classification_loss'], lv['regression_loss']], np.float64)
    out['train/grad_norms'] = np.array([float(cg.grad.double().norm()), float(rgg.grad.double().norm())])
    idx = cases.sample_index(cl.shape[1])
    out['train/grad_cls'], out['train/grad_reg'] = cg.grad[:, idx].numpy(), rgg.grad[:, idx].numpy()
    print('train', out['train/counts'], out['train/loss'], out['train/grad_norms'])
    np.savez_compressed(os.path.join(os.environ.get('LFD_GOLDEN_OUT', HERE), 'ref_fullsize_model.npz'), **out)


if __name__ == '__main__':
    main()
