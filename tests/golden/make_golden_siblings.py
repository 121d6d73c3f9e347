"""tests/golden/make_golden_siblings.py -- golden fixtures of the sibling meta-architectures, produced by
running the REAL reference classes (/root/reference: FCOS, LFDv2, FPN, SimpleFPN, FCOSHead, LFDHead with 3x3 convs),
imported through oracle/ref_import.py in the build container.

    python tests/golden/make_golden_siblings.py

Outputs (committed): ref_sibling_<NAME>.npz (forward tensors, get_results for two metas, get_loss values and prediction
gradients of the compositions in lfd_amd.configs.SIBLINGS), ref_sibling_necks.npz (FPN / SimpleFPN alone, the corner
options), ref_sibling_results.npz (get_results of both meta-architectures on large seeded prediction tensors).
Weights and inputs are NOT stored: they are regenerated from seeds (configs.synthetic_weights draws per state_dict key,
so the reference modules and this package's get the same tensors); the sha256 of the reference state_dict is stored.
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
sys.path.insert(0, HERE)
warnings.filterwarnings('ignore')

from oracle import ref_import  # noqa: E402
from lfd_amd import configs  # noqa: E402


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


from sibling_cases import (NECK_CASES, RESULT_CASES, neck_inputs, result_inputs, synth_annotations,  # noqa: E402
                           synth_annotations_overlapping)


def main():
    M = ref_import.import_reference()
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN

    # ---------------------------------------------------------------- 1. whole models
    for name, spec in configs.SIBLINGS.items():
        N_, H, W = 2, 128, 160
        model = configs.build_sibling(name, RB, RN, RH, M, RL, seed=1)
        mine = configs.build_sibling_model(name, seed=1)
        sd, sd_m = model.state_dict(), mine.state_dict()
        assert sorted(sd) == sorted(sd_m), (sorted(set(sd) ^ set(sd_m)))
        for k in sd:
            assert torch.equal(sd[k], sd_m[k]), k
        model.eval()
        x = torch.rand(N_, 3, H, W, generator=torch.Generator().manual_seed(7)) * 2 - 1
        with torch.no_grad():
            outs = model(x)
        nl = len(model._point_strides)
        sizes = [model.head_indexes_to_feature_map_sizes[i] for i in range(nl)]
        res = dict(x_seed=7, shape=np.array([N_, H, W]), sizes=np.array(sizes), sha=state_sha(sd), cls=outs[0].numpy(),
                   reg=outs[1].numpy())
        if len(outs) == 3:
            res['ctr'] = outs[2].numpy()
            sc = outs[0].sigmoid() * outs[2].sigmoid()
        elif spec.get('classification_loss_type') == 'CrossEntropyLoss':
            sc = outs[0].softmax(-1)[..., :-1]
        else:
            sc = outs[0].sigmoid()
        thr = float(np.quantile(sc.numpy(), 0.9))
        model._classification_threshold = thr
        model._nms_cfg = dict(type='nms', iou_thr=0.45)
        res['results_thr'], res['results_iou'] = thr, 0.45
        res['results'] = json.dumps(model.get_results(outs, [dict(resized_height=H, resized_width=W, resize_scale=1.0)] * N_))
        res['results_scaled'] = json.dumps(model.get_results(outs, [dict(resized_height=H - 6, resized_width=W - 10,
                                                                         resize_scale=0.5)] * N_))
        ann = synth_annotations(5, N_, H, W, spec['head']['num_classes'])
        preds = [o.clone().requires_grad_(True) for o in outs]
        lo = model.get_loss(tuple(preds), ann)
        lo['loss'].backward()
        res['loss_values'] = json.dumps(lo['loss_values'])
        for nm, p in zip(('dcls', 'dreg', 'dctr'), preds):
            res[nm] = p.grad.numpy()
        pts = model.generate_point_coordinates(model.head_indexes_to_feature_map_sizes)
        tg = model.annotation_to_target(pts, [torch.from_numpy(b) for b, _ in ann], [torch.from_numpy(l) for _, l in ann])
        res['cls_target'], res['reg_target'] = tg[0].numpy(), tg[1].numpy()
        np.savez_compressed(os.path.join(HERE, 'ref_sibling_%s.npz' % name), **res)
        print(name, 'P', outs[0].shape[1], 'results', [len(r) for r in json.loads(res['results'])], lo['loss_values'])

    # ---------------------------------------------------------------- 1b. FCOSv1: FCOS's network, multi-label targets / loss
    spec = dict(configs.SIBLINGS['FCOS_FPN'], meta='FCOSv1')
    model = configs.build_sibling(spec, RB, RN, RH, M, RL, seed=1)
    g = np.load(os.path.join(HERE, 'ref_sibling_FCOS_FPN.npz'))
    N_, H, W = [int(v) for v in g['shape']]
    for i, hw in enumerate(g['sizes'].tolist()):
        model._head_indexes_to_feature_map_sizes[i] = tuple(hw)
    preds = [torch.from_numpy(g[k]).requires_grad_(True) for k in ('cls', 'reg', 'ctr')]
    ann = synth_annotations_overlapping(5, N_, H, W, spec['head']['num_classes'])
    lo = model.get_loss(tuple(preds), ann)
    lo['loss'].backward()
    res = dict(loss_values=json.dumps(lo['loss_values']))
    for nm, p in zip(('dcls', 'dreg', 'dctr'), preds):
        res[nm] = p.grad.numpy()
    pts = model.generate_point_coordinates(model.head_indexes_to_feature_map_sizes)
    tg = model.annotation_to_target(pts, [torch.from_numpy(b) for b, _ in ann], [torch.from_numpy(l) for _, l in ann])
    res['cls_target'], res['reg_target'] = tg[0].numpy(), tg[1].numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_sibling_FCOSV1.npz'), **res)
    print('FCOSv1', lo['loss_values'])

    # ---------------------------------------------------------------- 2. necks alone
    out = {}
    for name, kind, kw in NECK_CASES:
        neck = getattr(RN, kind)(num_input_channels_list=[64, 64, 128], num_input_strides_list=[8, 16, 32], **kw)
        configs.synthetic_weights(neck, seed=3)
        neck.eval()
        with torch.no_grad():
            ys = neck([t.clone() for t in neck_inputs(name)])
        for i, y in enumerate(ys):
            out['%s_out%d' % (name, i)] = y.numpy()
        out['%s_strides' % name] = np.array(neck.num_output_strides_list)
        out['%s_sha' % name] = state_sha(neck.state_dict())
    np.savez_compressed(os.path.join(HERE, 'ref_sibling_necks.npz'), **out)

    # ---------------------------------------------------------------- 3. get_results on large seeded predictions
    out = {}
    for ci, case in enumerate(RESULT_CASES):
        cls, reg, ctr = result_inputs(case)
        if case['meta'] == 'FCOS':
            m = M.FCOS(num_classes=case['C'], regress_ranges=tuple((0, 1) for _ in case['sizes']), point_strides=case['strides'],
                       nms_threshold=case['iou'], pre_nms_bbox_limit=case['pre'], post_nms_bbox_limit=case['post'])
            sc = cls.sigmoid() * ctr.sigmoid()
            preds = (cls, reg, ctr)
        else:
            closs = RL.CrossEntropyLoss() if case['ce'] else RL.FocalLoss()
            rloss = getattr(RL, case['loss'])()
            m = M.LFDv2(num_classes=case['C'], regression_ranges=case['ranges'], point_strides=case['strides'],
                        classification_loss_func=closs, regression_loss_func=rloss, distance_to_bbox_mode=case['mode'],
                        nms_threshold=case['iou'], pre_nms_bbox_limit=case['pre'], post_nms_bbox_limit=case['post'])
            sc = cls.softmax(-1)[..., :-1] if case['ce'] else cls.sigmoid()
            preds = (cls, reg)
        for i, hw in enumerate(case['sizes']):
            m._head_indexes_to_feature_map_sizes[i] = hw
        thr = float(np.quantile(sc.numpy(), case['q']))
        m._classification_threshold = thr
        Hh, Ww = case['sizes'][0][0] * case['strides'][0], case['sizes'][0][1] * case['strides'][0]
        metas = [dict(resized_height=Hh, resized_width=Ww, resize_scale=1.0), dict(resized_height=Hh - 9, resized_width=Ww - 14,
                                                                                    resize_scale=0.75)]
        out['thr_%d' % ci] = thr
        out['results_%d' % ci] = json.dumps(m.get_results(preds, metas))
        print('results case', ci, [len(r) for r in json.loads(out['results_%d' % ci])])
    np.savez_compressed(os.path.join(HERE, 'ref_sibling_results.npz'), **out)


if __name__ == '__main__':
    main()
