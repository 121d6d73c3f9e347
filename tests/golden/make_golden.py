"""tests/golden/make_golden.py -- regenerates the golden fixtures by running the REAL reference
(/root/reference, imported through oracle/ref_import.py) in the build container.

    python tests/golden/make_golden.py

Outputs (committed): known_answers.json, ref_nms.npz, ref_nms_cpu_extra.npz, ref_multiclass_nms.npz,
ref_model_<ARCH>.npz.  The GPU box has no /root/reference: tests only read these files.
Model weights are NOT stored: they are regenerated from torch.manual_seed(666) + the
deterministic perturbation (lfd_amd.configs.perturb_weights); the sha256 of the reference
state_dict is stored so a drift in init order or RNG is detected instead of silently changing
the inputs.
"""
import hashlib
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
warnings.filterwarnings('ignore')

from oracle import build_ref, ref_import  # noqa: E402
from lfd_amd import configs  # noqa: E402  (only the arch dicts + perturbation helper)


def state_sha(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def synth_boxes(rng, k, W=1920, H=1080):
    """centres uniform over the frame, sizes logU[4,320], tie-free scores."""
    cx, cy = rng.uniform(0, W, k), rng.uniform(0, H, k)
    s = np.exp(rng.uniform(np.log(4), np.log(320), (k, 2)))
    b = np.stack([cx - s[:, 0] / 2, cy - s[:, 1] / 2, cx + s[:, 0] / 2, cy + s[:, 1] / 2], 1)
    b = b.clip(0, [W, H, W, H]).astype(np.float32)
    sc = (rng.permutation(k).astype(np.float32) + 1) / (k + 1)
    return b, sc.astype(np.float32)


def main():
    only_model = sys.argv[sys.argv.index('--only-model') + 1] if '--only-model' in sys.argv else None
    M = ref_import.import_reference()
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN
    from lfd.model.losses.iou_loss import bbox_overlaps
    from lfd.model.losses.utils import weighted_loss
    from lfd.model.utils import nms as ref_nms_py, soft_nms as ref_soft_nms_py, multiclass_nms as ref_mc_nms
    ext = build_ref.load_ref()

    # ---------------------------------------------------------------- 1. docstring known answers
    ka = {}
    d = np.array([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9], [49.2, 31.8, 51.0, 35.4, 0.5],
                  [35.1, 11.5, 39.1, 15.7, 0.5], [35.6, 11.8, 39.3, 14.2, 0.5], [35.3, 11.5, 39.9, 14.5, 0.4],
                  [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
    sup, inds = ref_nms_py(d, 0.6)                                   # nms.py:25-34
    assert len(inds) == len(sup) == 3
    ka['nms_docstring'] = dict(dets=d.tolist(), iou_thr=0.6, expected_len=3, keep=inds.tolist())
    d2 = np.array([[4., 3., 5., 3., 0.9], [4., 3., 5., 4., 0.9], [3., 1., 3., 1., 0.5], [3., 1., 3., 1., 0.5],
                   [3., 1., 3., 1., 0.4], [3., 1., 3., 1., 0.0]], dtype=np.float32)
    nd, ni = ref_soft_nms_py(d2, 0.6, sigma=0.5)                     # nms.py:80-88
    assert len(ni) == len(nd) == 5
    ka['soft_nms_docstring'] = dict(dets=d2.tolist(), iou_thr=0.6, sigma=0.5, method='linear', min_score=1e-3,
                                    expected_len=5, new_dets=nd.tolist(), inds=ni.tolist())
    b1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    ka['bbox_overlaps_docstring'] = dict(b1=b1.tolist(), b2=b2.tolist(),                 # iou_loss.py:28-42
                                         expected=[[0.5, 0, 0], [0, 0, 1.0], [0, 0, 0]],
                                         got=bbox_overlaps(b1, b2).tolist(),
                                         aligned=bbox_overlaps(b1, b2, is_aligned=True).tolist())

    @weighted_loss
    def l1_loss(pred, target):
        return (pred - target).abs()
    p, t, w = torch.Tensor([0, 2, 3]), torch.Tensor([1, 1, 1]), torch.Tensor([1, 0, 1])
    ka['weighted_loss_docstring'] = dict(pred=p.tolist(), target=t.tolist(), weight=w.tolist(),   # losses/utils.py:67-85
                                         mean=float(l1_loss(p, t)), weighted_mean=float(l1_loss(p, t, w)),
                                         none=l1_loss(p, t, reduction='none').tolist(),
                                         avg_factor_2=float(l1_loss(p, t, w, avg_factor=2)))
    if only_model is None:
        prev = json.load(open(os.path.join(HERE, 'known_answers.json')))
        ka.update({k: v for k, v in prev.items() if k not in ka})      # keys owned by other generators (reference_model_configs)
        json.dump(ka, open(os.path.join(HERE, 'known_answers.json'), 'w'), indent=1)

    # ---------------------------------------------------------------- 2. reference CPU nms_ext on seeded boxes
    rng = np.random.default_rng(1234)
    out = {}
    cases = []
    for ci, (k, thr) in enumerate([(1, 0.3), (2, 0.5), (7, 0.6), (63, 0.3), (64, 0.3), (65, 0.4), (129, 0.3),
                                   (300, 0.4), (1000, 0.3), (1000, 0.0), (257, 0.9)]):
        b, s = synth_boxes(rng, k, 640, 480) if k < 500 else synth_boxes(rng, k)
        dets = np.concatenate([b, s[:, None]], 1).astype(np.float32)
        keep = ext.nms(torch.from_numpy(dets), float(thr)).numpy()
        out['dets_%d' % ci], out['keep_%d' % ci] = dets, keep
        cases.append((k, thr))
    # degenerate boxes (zero area -> 0/0 = NaN -> never suppressed) and exact duplicates (tie-free scores)
    dets = np.array([[10, 10, 10, 10, .9], [10, 10, 10, 10, .8], [5, 5, 20, 20, .7], [5, 5, 20, 20, .6],
                     [0, 0, 1, 1, .5]], np.float32)
    out['dets_%d' % len(cases)] = dets
    out['keep_%d' % len(cases)] = ext.nms(torch.from_numpy(dets), 0.5).numpy()
    cases.append((5, 0.5))
    out['cases'] = np.array(cases, np.float64)
    if only_model is None:
        np.savez_compressed(os.path.join(HERE, 'ref_nms.npz'), **out)

    # ---------------------------------------------------------------- 2b. reference CPU soft_nms / nms_match (nms_cpu.cpp:76-283)
    out = {}
    ci = 0
    rng_x = np.random.default_rng(4321)     # own stream: this fixture can be regenerated alone
    for k, thr, method, sigma, min_score in [(6, 0.6, 1, 0.5, 1e-3), (50, 0.3, 1, 0.5, 0.05), (50, 0.3, 2, 0.5, 0.05), (300, 0.4, 2, 0.3, 0.1),
                                             (300, 0.5, 0, 0.5, 0.2), (1, 0.5, 1, 0.5, 1e-3)]:
        b, s = synth_boxes(rng_x, k, 640, 480)
        dets = np.concatenate([b, s[:, None]], 1).astype(np.float32)
        soft = ext.soft_nms(torch.from_numpy(dets), float(thr), int(method), float(sigma), float(min_score)).numpy()
        match = ext.nms_match(torch.from_numpy(dets), float(thr))
        out['dets_%d' % ci], out['params_%d' % ci], out['soft_%d' % ci] = dets, np.array([thr, method, sigma, min_score], np.float64), soft
        out['match_sizes_%d' % ci] = np.array([len(m) for m in match], np.int64)
        out['match_members_%d' % ci] = np.array([i for m in match for i in m], np.int64)
        ci += 1
    out['num_cases'] = np.array(ci)
    if only_model is None:
        np.savez_compressed(os.path.join(HERE, 'ref_nms_cpu_extra.npz'), **out)

    # ---------------------------------------------------------------- 3. reference python multiclass_nms
    out = {}
    mc = []
    for ci, (n, C, sthr, ithr, agn) in enumerate([(200, 1, 0.3, 0.4, False), (300, 5, 0.5, 0.3, False),
                                                  (300, 5, 0.5, 0.3, True), (150, 45, 0.9, 0.1, False),
                                                  (50, 3, 0.999, 0.5, False)]):
        b, _ = synth_boxes(rng, n, 1280, 720)
        sc = rng.uniform(0, 1, (n, C)).astype(np.float32)
        sc = np.concatenate([sc, np.zeros((n, 1), np.float32)], 1)
        cfg = dict(type='nms', iou_thr=ithr)
        if agn:
            cfg['class_agnostic'] = True
        dets, labels = ref_mc_nms(torch.from_numpy(b), torch.from_numpy(sc), sthr, cfg)
        out['boxes_%d' % ci], out['scores_%d' % ci] = b, sc
        out['dets_%d' % ci], out['labels_%d' % ci] = dets.numpy(), labels.numpy()
        mc.append((n, C, sthr, ithr, int(agn)))
    out['cases'] = np.array(mc, np.float64)
    if only_model is None:
        np.savez_compressed(os.path.join(HERE, 'ref_multiclass_nms.npz'), **out)

    # ---------------------------------------------------------------- 4. reference model runs
    for name, (N, H, W) in (('WIDERFACE_LFD_XS', (2, 96, 128)), ('WIDERFACE_LFD_S', (1, 72, 104)),
                            ('TT100K_LFD_L', (1, 64, 96)), ('TL_LFD_L', (1, 64, 128)), ('TL_LFD_S', (1, 72, 120))):
        if only_model is not None and name != only_model:
            continue
        arch = configs.ARCHS[name]
        model = configs.build_modules(arch, RB.LFDResNet, RN.SimpleNeck, RH.LFDHead, M.LFD, RL.FocalLoss,
                                      RL.IoULoss, RL.CrossEntropyLoss, seed=666, qfl_cls=RL.QualityFocalLoss)
        sha_init = state_sha(model.state_dict())
        configs.perturb_weights(model, seed=1)
        sha = state_sha(model.state_dict())
        model.eval()
        x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(7)) * 2 - 1
        with torch.no_grad():
            cls, reg = model(x)
        sizes = [model.head_indexes_to_feature_map_sizes[i] for i in range(len(arch['regression_ranges']))]
        res = dict(x_seed=7, shape=np.array([N, H, W]), cls=cls.numpy(), reg=reg.numpy(), sizes=np.array(sizes),
                   sha_init=sha_init, sha=sha)
        # get_results at a threshold giving ~15 % of the (point, class) scores as candidates
        ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
        sc = cls.softmax(-1)[..., :-1] if ce else cls.sigmoid()
        thr = float(np.quantile(sc.numpy(), 0.85))
        model._classification_threshold = thr
        model._nms_cfg = dict(type='nms', iou_thr=0.4)
        meta = [dict(resized_height=H, resized_width=W, resize_scale=1.0) for _ in range(N)]
        results = model.get_results((cls, reg), meta)
        res['results_thr'] = thr
        res['results_iou'] = 0.4
        res['results'] = json.dumps(results)
        meta2 = [dict(resized_height=H - 6, resized_width=W - 10, resize_scale=0.5) for _ in range(N)]
        res['results_scaled'] = json.dumps(model.get_results((cls, reg), meta2))
        # get_loss with synthetic annotations (xywh float32, labels int64)
        rs = np.random.default_rng(5)
        ann = []
        for _ in range(N):
            g = int(rs.integers(1, 6))
            wh = np.exp(rs.uniform(np.log(6), np.log(min(H, W) * 0.9), (g, 2)))
            xy = rs.uniform(0, [W, H], (g, 2)) - wh / 2
            bb = np.concatenate([xy, wh], 1).astype(np.float32)
            lb = rs.integers(0, arch['num_classes'], g).astype(np.int64)
            ann.append((bb, lb))
        model.train()      # get_loss itself is mode independent; forward again in eval for determinism
        model.eval()
        cls_g = cls.clone().requires_grad_(True)
        reg_g = reg.clone().requires_grad_(True)
        lo = model.get_loss((cls_g, reg_g), ann)
        lo['loss'].backward()
        pts = model.generate_point_coordinates(model.head_indexes_to_feature_map_sizes)
        ct, rt = model.annotation_to_target(pts, [torch.from_numpy(a[0]) for a in ann],
                                            [torch.from_numpy(a[1]) for a in ann])
        res['ann_boxes'] = np.concatenate([a[0] for a in ann], 0)
        res['ann_labels'] = np.concatenate([a[1] for a in ann], 0)
        res['ann_counts'] = np.array([len(a[1]) for a in ann])
        res['loss'] = np.array([lo['loss_values']['loss'], lo['loss_values']['classification_loss'],
                                lo['loss_values']['regression_loss']], np.float64)
        res['cls_targets'] = ct.numpy()
        res['reg_targets'] = rt.numpy()
        res['grad_cls'] = cls_g.grad.numpy()
        res['grad_reg'] = reg_g.grad.numpy()
        np.savez_compressed(os.path.join(HERE, 'ref_model_%s.npz' % name), **res)
        print(name, 'P', cls.shape[1], 'thr', thr, 'results', [len(r) for r in results], 'loss', res['loss'])


if __name__ == '__main__':
    main()
