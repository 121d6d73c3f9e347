"""LFDHead -- host-side mirror of lfd/model/head/lfd_head.py:30-185 (+ Scale :9-16).

Per level: optional shared merge path (k x (conv + norm + act)) or separate cls / reg towers,
then cls conv1x1 -> C (C+1 for CrossEntropyLoss :138-141) and reg conv1x1 -> 4, learnable
per-level Scale on reg for IoU-type losses (:64-65,179-180); weights optionally shared
across levels by aliasing head0's Sequentials (:67-82).  Child names, creation order and the
N(0, 0.01) init (:151-162) follow the reference so checkpoints and seeds line up.
"""
import torch
import torch.nn as nn

from ..backbone.lfd_resnet import build_activation, build_norm

__all__ = ['LFDHead', 'LFDHeadV1', 'Scale']

_UNION_LOSSES = ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss')


class Scale(nn.Module):
    def __init__(self, scale_factor=1.0):
        super().__init__()
        self._scale = nn.Parameter(torch.tensor(scale_factor, dtype=torch.float))

    def forward(self, x):
        return x * self._scale


class LFDHead(nn.Module):

    def __init__(self, num_classes, num_input_channels, num_heads, num_head_channels=128, num_conv_layers=2,
                 conv_kernel_size=1, activation_cfg=dict(type='ReLU', inplace=True),
                 norm_cfg=dict(type='BatchNorm2d'), classification_loss_type='SmoothL1Loss',
                 regression_loss_type='SmoothL1Loss', share_head_flag=False, merge_path_flag=False):
        super().__init__()
        assert classification_loss_type in ['BCEWithLogitsLoss', 'FocalLoss', 'CrossEntropyLoss', 'QualityFocalLoss']
        assert regression_loss_type in ['SmoothL1Loss', 'MSELoss'] + list(_UNION_LOSSES)
        assert conv_kernel_size in [1, 3]
        self._num_classes = num_classes
        self._num_input_channels = num_input_channels
        self._num_head_channels = num_head_channels
        self._num_conv_layers = num_conv_layers
        self._conv_kernel_size = conv_kernel_size
        self._activation_cfg, self._norm_cfg = activation_cfg, norm_cfg
        self._share_head_flag, self._merge_path_flag = share_head_flag, merge_path_flag
        self._num_heads = num_heads
        self._classification_loss_type = classification_loss_type
        self._regression_loss_type = regression_loss_type

        if regression_loss_type in _UNION_LOSSES:
            self._scales = nn.ModuleList([Scale(1.0) for _ in range(num_heads)])
        for i in range(num_heads):
            if i == 0 or not share_head_flag:
                paths = self._build_head()
            else:
                paths = tuple(getattr(self, 'head0_%s_path' % n) for n in ('classification', 'regression', 'merge'))
            for name, path in zip(('classification', 'regression', 'merge'), paths):
                setattr(self, 'head%d_%s_path' % (i, name), path)
        self._init_weights()

    @property
    def num_cls_channels(self):
        return self._num_classes + (1 if self._classification_loss_type == 'CrossEntropyLoss' else 0)

    def _tower_layer(self, cin):
        k = self._conv_kernel_size
        out = [nn.Conv2d(cin, self._num_head_channels, kernel_size=k, stride=1, padding=int(k / 2),
                         bias=self._norm_cfg is None)]
        if self._norm_cfg is not None:
            out.append(build_norm(self._norm_cfg, self._num_head_channels))
        out.append(build_activation(self._activation_cfg))
        return out

    def _build_head(self):
        """lfd_head.py:86-149 (creation order matters for seeded init)."""
        cls_path, reg_path, merge_path = [], [], []
        for l in range(self._num_conv_layers):
            cin = self._num_input_channels if l == 0 else self._num_head_channels
            if self._merge_path_flag:
                merge_path += self._tower_layer(cin)
            else:
                cls_path += self._tower_layer(cin)
                reg_path += self._tower_layer(cin)
        cls_path.append(nn.Conv2d(self._num_head_channels, self.num_cls_channels, kernel_size=1, bias=True))
        reg_path.append(nn.Conv2d(self._num_head_channels, 4, kernel_size=1, bias=True))
        return nn.Sequential(*cls_path), nn.Sequential(*reg_path), nn.Sequential(*merge_path)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        raise RuntimeError('LFDHead is executed inside the fused LFD engine plan; call LFD.forward')


class LFDHeadV1(LFDHead):
    """lfd_head.py:188-343: LFDHead whose towers are always 1x1 and end WITHOUT the output convs -- those are per-level
    members `_classifiers[i]` / `_regressors[i]` even when the towers are shared; kaiming-normal init for every conv
    (:310-320).  Executed layer by layer on the gfx950 kernels (engine_sibling)."""

    def __init__(self, num_classes, num_input_channels, num_heads, num_head_channels=128, num_conv_layers=2,
                 activation_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='BatchNorm2d'),
                 classification_loss_type='SmoothL1Loss', regression_loss_type='SmoothL1Loss', share_head_flag=False,
                 merge_path_flag=False):
        assert classification_loss_type in ['BCEWithLogitsLoss', 'FocalLoss', 'CrossEntropyLoss']
        nn.Module.__init__(self)
        self._num_classes = num_classes
        self._num_input_channels = num_input_channels
        self._num_head_channels = num_head_channels
        self._num_conv_layers = num_conv_layers
        self._conv_kernel_size = 1
        self._activation_cfg, self._norm_cfg = activation_cfg, norm_cfg
        self._share_head_flag, self._merge_path_flag = share_head_flag, merge_path_flag
        self._num_heads = num_heads
        self._classification_loss_type = classification_loss_type
        self._regression_loss_type = regression_loss_type
        assert regression_loss_type in ['SmoothL1Loss', 'MSELoss'] + list(_UNION_LOSSES)
        if regression_loss_type in _UNION_LOSSES:
            self._scales = nn.ModuleList([Scale(1.0) for _ in range(num_heads)])
        self._classifiers = nn.ModuleList()
        self._regressors = nn.ModuleList()
        for i in range(num_heads):
            self._classifiers.append(nn.Conv2d(num_head_channels, self.num_cls_channels, kernel_size=1, bias=True))
            self._regressors.append(nn.Conv2d(num_head_channels, 4, kernel_size=1, bias=True))
            if i == 0 or not share_head_flag:
                paths = self._build_head()
            else:
                paths = tuple(getattr(self, 'head0_%s_path' % n) for n in ('classification', 'regression', 'merge'))
            for name, path in zip(('classification', 'regression', 'merge'), paths):
                setattr(self, 'head%d_%s_path' % (i, name), path)
        self._init_weights()

    def _build_head(self):
        cls_path, reg_path, merge_path = [], [], []
        for l in range(self._num_conv_layers):
            cin = self._num_input_channels if l == 0 else self._num_head_channels
            if self._merge_path_flag:
                merge_path += self._tower_layer(cin)
            else:
                cls_path += self._tower_layer(cin)
                reg_path += self._tower_layer(cin)
        return nn.Sequential(*cls_path), nn.Sequential(*reg_path), nn.Sequential(*merge_path)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
