"""tests/golden/train_step_cases.py -- the seeded inputs of the training-iteration fixtures, shared by the generator
(make_golden_train_step.py, runs the reference) and the tests (tests/test_train_golden.py, run this repository)."""
import numpy as np
import torch

# configuration -> (images, height, width): a focal + IoU loss model with a shared head, the 45-class cross-entropy one with
# separate towers, the 32-channel-stem model, and a TrafficLight one (quality focal loss, norm-free head: the documented
# training fallback) (WIDERFACE_LFD_S.py / TT100K_LFD_L.py / WIDERFACE_LFD_XS.py / TL_LFD_L.py)
CASES = {'WIDERFACE_LFD_S': (2, 128, 160), 'TT100K_LFD_L': (2, 96, 128), 'WIDERFACE_LFD_XS': (2, 96, 128), 'TL_LFD_L': (2, 64, 128)}
# WIDERFACE_LFD_S.py:217-241: SGD momentum 0.9, weight decay 1e-4, lr 0.1 x warm-up ratio 0.1 in the first iteration,
# clip_grad_norm_(max_norm=10, norm_type=2) during the first 5 epochs
LR, MOMENTUM, WEIGHT_DECAY = 0.01, 0.9, 1e-4
GRAD_CLIP = dict(max_norm=10, norm_type=2, duration=5)
ITERATIONS = 3
# round 4: the same protocol at sizes where EVERY BatchNorm of the network sees >= 512 elements per channel (the stride-64 maps
# of the cases above are 2 x 2..3 pixels x 2 images = 8-12 elements: batch statistics over a dozen fp16 values).  key -> (arch,
# images, height, width); the fixtures hold the iteration summaries only (losses, gradient norms, gradient / state summaries)
LARGE_CASES = {'WIDERFACE_LFD_S@8x512x512': ('WIDERFACE_LFD_S', 8, 512, 512), 'WIDERFACE_LFD_XS@8x512x512': ('WIDERFACE_LFD_XS', 8, 512, 512)}


def shape_of(name):
    """-> (arch name, images, height, width) for a key of CASES or LARGE_CASES"""
    if name in CASES:
        return (name,) + tuple(CASES[name])
    return LARGE_CASES[name]


def file_tag(name):
    return name.replace('@', '_at_')


def images(name):
    _, n, h, w = shape_of(name)
    return torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(11)) * 2 - 1


def annotations(name, num_classes):
    """per image (boxes xywh float32 [g, 4], labels int64 [g]), 1-5 boxes each"""
    _, n, h, w = shape_of(name)
    rs = np.random.default_rng(5)
    ann = []
    for _ in range(n):
        g = int(rs.integers(1, 6))
        wh = np.exp(rs.uniform(np.log(6), np.log(min(h, w) * 0.9), (g, 2)))
        xy = rs.uniform(0, [w, h], (g, 2)) - wh / 2
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), rs.integers(0, num_classes, g).astype(np.int64)))
    return ann
