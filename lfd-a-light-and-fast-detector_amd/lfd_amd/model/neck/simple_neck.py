"""SimpleNeck -- host-side mirror of lfd/model/neck/simple_neck.py:18-74.

Per level: conv1x1 (C_in -> num_neck_channels) + norm + activation, no top-down path.
Children named `neck{i}` = Sequential(conv, norm, act) exactly as the reference (:35-47) so
state_dict keys match.  Arithmetic runs in the gfx950 engine (the neck 1x1 is chained in
registers into the head's first 1x1 -- see csrc/head.hip).
"""
import torch.nn as nn

from ..backbone.lfd_resnet import build_activation, build_norm

__all__ = ['SimpleNeck']


class SimpleNeck(nn.Module):

    def __init__(self, num_neck_channels, num_input_channels_list, num_input_strides_list,
                 norm_cfg=dict(type='BatchNorm2d'), activation_cfg=dict(type='ReLU', inplace=True)):
        super().__init__()
        assert len(num_input_channels_list) == len(num_input_strides_list)
        self._num_neck_channels = num_neck_channels
        self._num_input_channels_list = num_input_channels_list
        self._num_input_strides_list = num_input_strides_list
        self._norm_cfg, self._activation_cfg = norm_cfg, activation_cfg
        self._num_inputs = len(num_input_channels_list)
        for level, cin in enumerate(num_input_channels_list):
            self.add_module('neck%d' % level, self._level(cin))
        self._init_weights()

    def _level(self, cin):
        """conv1x1 (bias only without a norm) -> norm -> activation, as one nn.Sequential (keys `neck{i}.0.weight`, ...)"""
        mods = [nn.Conv2d(cin, self._num_neck_channels, kernel_size=1, stride=1, padding=0, bias=self._norm_cfg is None)]
        if self._norm_cfg is not None:
            mods.append(build_norm(self._norm_cfg, self._num_neck_channels))
        mods.append(build_activation(self._activation_cfg))
        return nn.Sequential(*mods)

    def _init_weights(self):
        """simple_neck.py:51-61: kaiming_normal_(fan_out, relu) convs with zero bias, (1, 0) norms."""
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                for t, v in ((m.weight, 1), (m.bias, 0)):
                    if t is not None:
                        nn.init.constant_(t, v)
            elif isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    @property
    def num_output_strides_list(self):
        return self._num_input_strides_list

    def forward(self, inputs):
        raise RuntimeError('SimpleNeck is executed inside the fused LFD engine plan; call LFD.forward')
