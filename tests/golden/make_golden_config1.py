"""tests/golden/make_golden_config1.py -- BASELINE config 1: the reference's own `predict_for_single_image` (lfd.py:544-655, the
call WIDERFACE_train/predict.py:22 makes) on one 640 x 480 frame, WIDERFACE_LFD_XS, on the reference's CPU PyTorch path.

    python tests/golden/make_golden_config1.py

SURVEY 8d config 1: uint8 BGR frame `np.random.default_rng(0).integers(0, 256, (480, 640, 3))`, pre-processing = the
reference's simple_normalize `(x / 255 - 0.5) / 0.5` (augmentation_pipeline.py:31-36), thresholds of predict.py:22
(classification 0.5, NMS IoU 0.3) and, because seeded weights do not put many scores above 0.5, a second run at the
threshold that makes 10 % of the points candidates.  predict_for_single_image moves the batch and the model with `.cuda()`
(lfd.py:567-568): in this GPU-less container both calls are patched to the identity for the duration of the run -- the
arithmetic is the reference's, on CPU tensors.  Weights: seed 666 + configs.perturb_weights(seed=1), as everywhere.
Output: ref_config1_predict.npz (thresholds, result rows [label, score, x1, y1, w, h], the frame's sha256)."""
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

from oracle import ref_import  # noqa: E402
from lfd_amd import configs  # noqa: E402  (only the arch dicts + perturbation helper)


def frame():
    return np.random.default_rng(0).integers(0, 256, (480, 640, 3)).astype(np.uint8)


def simple_normalize(sample):
    """augmentation_pipeline.py:31-36"""
    sample['image'] = (sample['image'].astype(np.float32) / 255 - 0.5) / 0.5
    return sample


def main():
    M = ref_import.import_reference()
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN
    name = 'WIDERFACE_LFD_XS'
    model = configs.build_modules(configs.ARCHS[name], RB.LFDResNet, RN.SimpleNeck, RH.LFDHead, M.LFD, RL.FocalLoss,
                                  RL.IoULoss, RL.CrossEntropyLoss, seed=666, qfl_cls=RL.QualityFocalLoss)
    configs.perturb_weights(model, seed=1)
    img = frame()
    tensor_cuda, module_cuda = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        model.eval()
        with torch.no_grad():
            x = torch.from_numpy(simple_normalize({'image': img})['image'][None].transpose(0, 3, 1, 2))
            cls, _ = model(x)
        thr10 = float(np.quantile(cls.sigmoid().numpy(), 0.9))
        out = dict(frame_sha=hashlib.sha256(img.tobytes()).hexdigest())
        for tag, thr, iou in (('predict_py', 0.5, 0.3), ('q90', thr10, 0.3)):
            model._nms_cfg = dict(type='nms', iou_thr=0.5)       # LFD.__init__'s default; nms_threshold below overrides it (:630-631)
            res = model.predict_for_single_image(img, simple_normalize, classification_threshold=thr, nms_threshold=iou)
            out[tag + '/thr'], out[tag + '/iou'] = np.float64(thr), np.float64(iou)
            out[tag + '/results'] = json.dumps(res)
            print(tag, 'thr', thr, 'detections', len(res))
        out['sizes'] = np.array([model.head_indexes_to_feature_map_sizes[i] for i in range(len(model._point_strides))])
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = tensor_cuda, module_cuda
    np.savez_compressed(os.path.join(os.environ.get('LFD_GOLDEN_OUT', HERE), 'ref_config1_predict.npz'), **out)


if __name__ == '__main__':
    main()
