"""tests/golden/make_golden_box_losses.py -- golden vectors of the GIoU / DIoU / CIoU losses from the REAL reference
(lfd/model/losses/iou_loss.py:127-430, imported from /root/reference through oracle/ref_import.py; build container only):

    python tests/golden/make_golden_box_losses.py   ->  tests/golden/ref_box_losses.npz

Also SmoothL1Loss (beta 1 and 0.11) / L1Loss / MSELoss elementwise on the same pairs scaled by 1/100.
Per kind: the per-pair loss of the reference module (reduction 'none', eps 1e-6, fp32) and d(sum of losses)/d(pred) from
the reference's own autograd graph, for 768 seeded pairs (overlapping, disjoint, nested, distant).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

from oracle import ref_import  # noqa: E402


def pairs(seed=7, n=768):
    rng = np.random.default_rng(seed)
    c = rng.uniform(20, 400, (n, 2))
    s = np.exp(rng.uniform(np.log(4), np.log(200), (n, 2)))
    tgt = np.concatenate([c - s / 2, c + s / 2], 1)
    shift = rng.normal(0, 1, (n, 2)) * s * rng.choice([0.05, 0.5, 3.0], (n, 1))
    ps = s * np.exp(rng.normal(0, 0.5, (n, 2)))
    pred = np.concatenate([c + shift - ps / 2, c + shift + ps / 2], 1)
    pred[:32, :2] = tgt[:32, :2] + 1.0
    pred[:32, 2:] = tgt[:32, 2:] - 1.0          # nested
    return pred.astype(np.float32), tgt.astype(np.float32)


def main():
    ref_import.import_reference()
    import lfd.model.losses as RL
    pred, tgt = pairs()
    out = dict(pred=pred, target=tgt, eps=np.float32(1e-6))
    for kind, cls in (('giou', RL.GIoULoss), ('diou', RL.DIoULoss), ('ciou', RL.CIoULoss)):
        p = torch.from_numpy(pred).clone().requires_grad_(True)
        loss = cls(eps=1e-6, reduction='none', loss_weight=1.0)(p, torch.from_numpy(tgt))
        loss.sum().backward()
        out['loss_' + kind] = loss.detach().numpy()
        out['grad_' + kind] = p.grad.numpy()
    # LFD's "independent" regression losses (smooth_l1_loss.py, mse_loss.py) on the same pairs scaled to O(1) distances
    a, b = (pred / 100.0).astype(np.float32), (tgt / 100.0).astype(np.float32)
    b[:16] = a[:16]                                   # exact zeros: |x| derivative at 0
    out['pw_pred'], out['pw_target'] = a, b
    for kind, mod in (('smooth_l1', RL.SmoothL1Loss(beta=1.0, reduction='none')),
                      ('smooth_l1_b011', RL.SmoothL1Loss(beta=0.11, reduction='none')),
                      ('l1', RL.L1Loss(reduction='none')), ('mse', RL.MSELoss(reduction='none'))):
        p = torch.from_numpy(a).clone().requires_grad_(True)
        loss = mod(p, torch.from_numpy(b))
        loss.sum().backward()
        out['loss_' + kind] = loss.detach().numpy()
        out['grad_' + kind] = p.grad.numpy()
    # the two other classification losses LFD accepts: BCEWithLogitsLoss (float targets, as LFD.get_loss passes them, and
    # 1-based integer labels with per-row weights) and QualityFocalLoss (beta 2, 6 classes, label 6 = background)
    rng = np.random.default_rng(11)
    x = rng.normal(0, 2.5, (640, 6)).astype(np.float32)
    soft = np.where(rng.random((640, 6)) < 0.2, rng.random((640, 6)), 0).astype(np.float32)
    lab1 = rng.integers(0, 7, 640).astype(np.int64)            # BCE label path: 0 = background, k -> channel k - 1
    roww = rng.uniform(0.2, 2.0, 640).astype(np.float32)
    qlab = rng.integers(0, 7, 640).astype(np.int64)            # QFL: 6 = background
    qsc = rng.uniform(0.05, 1.0, 640).astype(np.float32)
    out.update(cls_logits=x, bce_soft=soft, bce_labels=lab1, bce_row_weight=roww, qfl_labels=qlab, qfl_scores=qsc)

    def run(key, fn):
        p = torch.from_numpy(x).clone().requires_grad_(True)
        loss = fn(p)
        loss.sum().backward()
        out['loss_' + key], out['grad_' + key] = loss.detach().numpy(), p.grad.numpy()
    run('bce_soft', lambda p: RL.BCEWithLogitsLoss(reduction='none')(p, torch.from_numpy(soft)))
    run('bce_labels', lambda p: RL.BCEWithLogitsLoss(reduction='none')(p, torch.from_numpy(lab1), weight=torch.from_numpy(roww)))
    run('qfl', lambda p: RL.QualityFocalLoss(beta=2.0, reduction='none')(p, (torch.from_numpy(qlab), torch.from_numpy(qsc))))
    np.savez_compressed(os.path.join(HERE, 'ref_box_losses.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
