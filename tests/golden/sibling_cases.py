"""tests/golden/sibling_cases.py -- the seeded inputs and case tables shared by the fixture generator
(make_golden_siblings.py, build container, runs the reference) and the tests that read the fixtures (CPU suite and GPU box):
nothing here touches /root/reference."""
import numpy as np
import torch


def synth_annotations(seed, n, H, W, num_classes):
    rs = np.random.default_rng(seed)
    ann = []
    for _ in range(n):
        g = int(rs.integers(2, 7))
        wh = np.exp(rs.uniform(np.log(8), np.log(min(H, W) * 0.9), (g, 2)))
        xy = rs.uniform(0, [W, H], (g, 2)) - wh / 2
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), rs.integers(0, num_classes, g).astype(np.int64)))
    return ann


def synth_annotations_overlapping(seed, n, H, W, num_classes):
    """synth_annotations + per image two nested boxes of different classes whose sizes share a regress range, so that some
    points lie strictly inside both: FCOSv1 marks both classes there, FCOS only the smaller box's"""
    ann = synth_annotations(seed, n, H, W, num_classes)
    out = []
    for i, (b, l) in enumerate(ann):
        x0, y0 = 20.0 + 10 * i, 16.0 + 6 * i
        extra = np.array([[x0, y0, 56.0, 48.0], [x0 + 5, y0 + 4, 46.0, 40.0]], np.float32)
        out.append((np.concatenate([b, extra]), np.concatenate([l, np.array([0, 1 % num_classes], np.int64)])))
    return out


NECK_CASES = [
    # (name, class, kwargs): inputs are 3 maps of 64 / 64 / 128 channels at 12x16, 6x8, 3x4 (seeded)
    ('fpn_pool_on_input', 'FPN', dict(num_output_channels=64, num_outputs=5, extra_on_input=True, extra_type='pooling',
                                      norm_on_lateral=False, relu_on_lateral=False, relu_before_extra=True, norm_cfg=None)),
    ('fpn_conv_on_input_gn', 'FPN', dict(num_output_channels=64, num_outputs=4, extra_on_input=True, extra_type='conv',
                                         norm_on_lateral=True, relu_on_lateral=True, relu_before_extra=False,
                                         norm_cfg=dict(type='GroupNorm', num_groups=8))),
    ('fpn_fewer_outputs', 'FPN', dict(num_output_channels=128, num_outputs=2, norm_cfg=None)),
    ('sfpn_neighbouring', 'SimpleFPN', dict(num_output_channels=64, num_outputs=5, extra_type='conv', relu_before_extra=True,
                                            neighbouring_mode=True)),
    ('sfpn_odd_sizes', 'SimpleFPN', dict(num_output_channels=128, num_outputs=3, norm_on_lateral=True,
                                         norm_cfg=dict(type='BatchNorm2d'))),
]
NECK_SHAPES = {'sfpn_odd_sizes': [(13, 17), (7, 9), (4, 5)]}          # nearest upsampling at non-integer ratios


def neck_inputs(name, seed=11):
    g = torch.Generator().manual_seed(seed)
    shapes = NECK_SHAPES.get(name, [(12, 16), (6, 8), (3, 4)])
    return [torch.rand(1, c, h, w, generator=g) * 2 - 0.5 for c, (h, w) in zip((64, 64, 128), shapes)]


RESULT_CASES = [
    # (meta, C, sizes, strides, ranges, ce, mode / loss type, thr quantile, iou, pre, post, seed)
    dict(meta='FCOS', C=3, sizes=[(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], strides=[8, 16, 32, 64, 128], pre=300, post=100,
         q=0.97, iou=0.5, seed=21),
    dict(meta='FCOS', C=1, sizes=[(50, 70), (25, 35), (13, 18)], strides=[8, 16, 32], pre=1000, post=-1, q=0.9, iou=0.4, seed=22),
    dict(meta='LFDv2', C=4, sizes=[(64, 96), (32, 48), (16, 24), (8, 12)], strides=[4, 8, 16, 32], pre=500, post=50, q=0.98,
         iou=0.45, seed=23, ce=False, mode='exp', loss='IoULoss', ranges=((4, 32), (32, 64), (64, 128), (128, 256))),
    dict(meta='LFDv2', C=6, sizes=[(40, 64), (20, 32), (10, 16)], strides=[8, 16, 32], pre=200, post=40, q=0.95, iou=0.5, seed=24,
         ce=True, mode='sigmoid', loss='IoULoss', ranges=((10, 40), (40, 80), (80, 160))),
    dict(meta='LFDv2', C=2, sizes=[(30, 40), (15, 20)], strides=[8, 16], pre=100, post=-1, q=0.9, iou=0.3, seed=25, ce=False,
         mode='exp', loss='SmoothL1Loss', ranges=((8, 64), (64, 256))),
]


def result_inputs(case):
    """seeded prediction tensors of a RESULT_CASES entry: (cls [N,P,C'], reg [N,P,4], ctr [N,P,1] | None), N = 2"""
    g = torch.Generator().manual_seed(case['seed'])
    P = sum(h * w for h, w in case['sizes'])
    Cc = case['C'] + (1 if case.get('ce') else 0)
    cls = torch.randn(2, P, Cc, generator=g) * 2.0 - 1.0
    if case['meta'] == 'FCOS':
        reg = torch.rand(2, P, 4, generator=g) * 60.0 + 1.0          # distances (the head's exp output)
        ctr = torch.randn(2, P, 1, generator=g) * 1.5
        return cls, reg, ctr
    if case['loss'] == 'SmoothL1Loss':
        reg = torch.rand(2, P, 4, generator=g) * 0.6                 # fractions of the range
    else:
        reg = torch.randn(2, P, 4, generator=g) * 1.2 + (2.0 if case['mode'] == 'exp' else 0.0)
    return cls, reg, None
