"""tests/golden/make_golden_fullsize_results.py -- the REAL reference's get_results on the full point grids of BASELINE
configs 2 / 3 / 4 (WIDERFACE_LFD_S 1080p: 43,620 points; WIDERFACE_LFD_L 4K: 690,600; TT100K_LFD_L 720p: 76,520 x 45 classes).

    python tests/golden/make_golden_fullsize_results.py

The small fixtures of make_golden.py pin decode / multiclass_nms / batched_nms' class-offset trick / result packing on a few
hundred points; the index arithmetic over the five levels, the class offsets `label * (max(bboxes) + 1)` (nms.py:119-158)
at 4K coordinates and 45 classes, and K = 256 / 4096 candidates only happen at the real grids.  Logits are synthetic
(fullsize_cases.logits: regenerated from a numpy seed by the tests, not stored), the thresholds are the quantiles giving
K candidates; stored: the thresholds and the reference's result rows [label | score, x1, y1, w, h] (lfd.py:418-431).
Output: ref_fullsize_results.npz.  Consumer: tests/test_oracle_golden.py (the CPU oracle; the device path is compared
with that oracle bit for bit at the same grids in tests/test_gpu_parity_fullsize.py)."""
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
from lfd_amd import configs  # noqa: E402  (only the arch dicts)
import fullsize_cases as cases  # noqa: E402


def main():
    M = ref_import.import_reference()
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN
    out = {}
    for key, (name, (H, W), sizes) in cases.GRIDS.items():
        arch = configs.ARCHS[name]
        model = configs.build_modules(arch, RB.LFDResNet, RN.SimpleNeck, RH.LFDHead, M.LFD, RL.FocalLoss,
                                      RL.IoULoss, RL.CrossEntropyLoss, seed=666, qfl_cls=RL.QualityFocalLoss)
        model.eval()
        ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
        channels = arch['num_classes'] + (1 if ce else 0)
        cls, reg = cases.logits(key, channels)
        for i, s in enumerate(sizes):                       # what LFD.forward records for this frame size (lfd.py:532)
            model._head_indexes_to_feature_map_sizes[i] = tuple(s)
        tc, tr = torch.from_numpy(cls), torch.from_numpy(reg)
        sc = tc.softmax(-1)[..., :-1] if ce else tc.sigmoid()                      # lfd.py:449-452
        flat = np.sort(sc.numpy().reshape(-1))
        for si, (K, iou, agn, scale) in enumerate(cases.SETTINGS):
            thr = float((flat[-K] + flat[-K - 1]) / 2)                              # between two scores: exactly K candidates
            model._classification_threshold = thr
            model._nms_cfg = dict(type='nms', iou_thr=iou)
            if agn:
                model._nms_cfg['class_agnostic'] = True
            meta = [dict(resized_height=H, resized_width=W, resize_scale=scale)]
            res = model.get_results((tc, tr), meta)                                 # lfd.py:434-509, nms.py:161-220
            out['%s/thr_%d' % (key, si)] = np.float64(thr)
            rows = np.array(res[0], np.float64).reshape(-1, 6)
            assert np.array_equal(rows[:, 1:].astype(np.float32).astype(np.float64), rows[:, 1:])      # fp32 values: stored losslessly
            out['%s/labels_%d' % (key, si)] = rows[:, 0].astype(np.int16)
            out['%s/rows_%d' % (key, si)] = rows[:, 1:].astype(np.float32)                             # score, x1, y1, w, h
            print(key, name, 'P', cls.shape[1], 'K', K, 'iou', iou, 'agnostic', agn, 'kept', len(res[0]))
    np.savez_compressed(os.path.join(os.environ.get('LFD_GOLDEN_OUT', HERE), 'ref_fullsize_results.npz'), **out)


if __name__ == '__main__':
    main()
