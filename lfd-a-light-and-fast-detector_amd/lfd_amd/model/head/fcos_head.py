"""FCOSHead -- host-side mirror of lfd/model/head/fcos_head.py:20-154.

Two towers shared by all levels (num_layers x [conv3x3, (norm), ReLU]); 3x3 output convs: classification (C channels) and
centerness (1) on the classification tower, regression (4) on the regression tower, the regression output going through
the level's learnable Scale and exp (:145-146).  Member names follow the reference (`_classification_path.{k}`,
`_regression_path.{k}`, `_classification`, `_centerness`, `_regression`, `_scales.{i}._scale`) so state_dict keys match;
init: N(0, 0.01) weights, zero biases, classification bias = -log((1 - 0.01) / 0.01) (:83-127).

Execution: inside FCOS.forward the whole network runs on the gfx950 engine (..engine_sibling); under autograd the children
run as PyTorch-ROCm modules on the device (training-only route); CPU tensors are refused.
"""
import math

import torch.nn as nn

from ... import _lib
from ..backbone.lfd_resnet import build_norm
from .lfd_head import Scale

__all__ = ['FCOSHead']


class FCOSHead(nn.Module):

    def __init__(self, num_classes, num_input_channels, num_head_channels=256, num_heads=5, num_layers=4, norm_cfg=None):
        super().__init__()
        if norm_cfg is not None:
            assert isinstance(norm_cfg, dict) and norm_cfg.get('type') in ['BatchNorm2d', 'GroupNorm']
            if norm_cfg['type'] == 'GroupNorm':
                assert 'num_groups' in norm_cfg
        self._num_classes = num_classes
        self._num_input_channels = num_input_channels
        self._num_head_channels = num_head_channels
        self._num_heads = num_heads
        self._num_layers = num_layers
        self._norm_cfg = norm_cfg
        ch = num_head_channels
        self._classification_path = nn.ModuleList()
        self._regression_path = nn.ModuleList()
        for layer in range(num_layers):
            cin = num_input_channels if layer == 0 else ch
            # creation order as in the reference (classification conv, its norm, regression conv, its norm, the two ReLUs)
            for path in (self._classification_path, self._regression_path):
                path.append(nn.Conv2d(cin, ch, kernel_size=3, stride=1, padding=1, bias=norm_cfg is None))
                if norm_cfg is not None:
                    path.append(build_norm(norm_cfg, ch))
            self._classification_path.append(nn.ReLU(inplace=True))
            self._regression_path.append(nn.ReLU(inplace=True))
        self._classification = nn.Conv2d(ch, num_classes, kernel_size=3, stride=1, padding=1, bias=True)
        self._centerness = nn.Conv2d(ch, 1, kernel_size=3, stride=1, padding=1, bias=True)
        self._regression = nn.Conv2d(ch, 4, kernel_size=3, stride=1, padding=1, bias=True)
        self._scales = nn.ModuleList([Scale(1.0) for _ in range(num_heads)])
        self._init_weights()

    def _init_weights(self):
        for path in (self._classification_path, self._regression_path):
            for m in path:
                if isinstance(m, nn.Conv2d):
                    nn.init.normal_(m.weight, mean=0, std=0.01)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
        prior = 0.01
        for conv, bias in ((self._classification, -math.log((1 - prior) / prior)), (self._regression, 0.),
                           (self._centerness, 0.)):
            nn.init.normal_(conv.weight, mean=0, std=0.01)
            nn.init.constant_(conv.bias, bias)

    def forward(self, inputs):
        """training route only (autograd over PyTorch-ROCm modules); inference goes through FCOS.forward -> engine"""
        assert isinstance(inputs, (list, tuple)) and len(inputs) == self._num_heads
        _lib.require_cuda(inputs[0], 'FCOSHead.forward')
        cls_out, reg_out, ctr_out = [], [], []
        for i, x in enumerate(inputs):
            c, r = x, x
            for m in self._classification_path:
                c = m(c)
            for m in self._regression_path:
                r = m(r)
            cls_out.append(self._classification(c))
            ctr_out.append(self._centerness(c))
            reg_out.append(self._scales[i](self._regression(r)).float().exp())
        return cls_out, reg_out, ctr_out
