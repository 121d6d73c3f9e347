"""FPN and the machinery it shares with SimpleFPN -- host-side mirror of lfd/model/neck/fpn.py:17-152.

Lateral 1x1 convs (optionally + norm, + ReLU), top-down `lateral[i-1] += nearest_upsample(lateral[i])`, then one output
path per level: a 3x3 conv for the pyramid levels, [ReLU] + (3x3 stride-2 conv | MaxPool 3/2/1) for the extra levels, fed
from the last input (`extra_on_input`) or the previous output.  Children are named `lateral{i}` / `fpn_out{i}` and are
nn.Sequential with the reference's member order, so state_dict keys match (`lateral0.0.weight`, `fpn_out5.1.bias`, ...).

Inference arithmetic runs on the gfx950 kernels (..engine_sibling: conv / upsample-add / relu / maxpool behind the C ABI).
Inside a meta-architecture the whole network goes through the engine at once; called stand-alone in eval mode the neck
converts its NCHW fp32 inputs, runs the same kernels and returns NCHW fp32.  Under autograd (training) the children run as
PyTorch-ROCm modules on the device -- the training-only route the LFD class documents as well; CPU tensors are refused.
"""
import torch
import torch.nn as nn

from ... import _lib
from ..backbone.lfd_resnet import build_norm

__all__ = ['FPN']


class _PyramidNeck(nn.Module):
    """what FPN and SimpleFPN have in common; subclasses say what a pyramid level's output path is and how convs start"""

    def __init__(self, num_input_channels_list, num_input_strides_list, num_output_channels, num_outputs, extra_on_input,
                 extra_type, norm_on_lateral, relu_on_lateral, relu_before_extra, norm_cfg):
        super().__init__()
        assert num_outputs >= 1
        assert extra_type in ['conv', 'pooling']
        if norm_on_lateral:
            assert norm_cfg is not None
        if norm_cfg is not None:
            assert norm_cfg.get('type') in ['BatchNorm2d', 'GroupNorm']
            if norm_cfg['type'] == 'GroupNorm':
                assert 'num_groups' in norm_cfg
        assert len(num_input_channels_list) == len(num_input_strides_list)
        self._num_input_channels_list = num_input_channels_list
        self._num_input_strides_list = num_input_strides_list
        self._num_inputs = len(num_input_channels_list)
        self._num_output_channels = num_output_channels
        self._num_outputs = num_outputs
        self._extra_on_input = extra_on_input
        self._extra_type = extra_type
        self._norm_on_lateral = norm_on_lateral
        self._relu_on_lateral = relu_on_lateral
        self._relu_before_extra = relu_before_extra
        self._norm_cfg = norm_cfg

    def _build(self):
        co = self._num_output_channels
        for i, cin in enumerate(self._num_input_channels_list):
            mods = [nn.Conv2d(cin, co, kernel_size=1, stride=1, padding=0, bias=not self._norm_on_lateral)]
            if self._norm_on_lateral:
                mods.append(build_norm(self._norm_cfg, co))
            if self._relu_on_lateral:
                mods.append(nn.ReLU(inplace=False))
            self.add_module('lateral%d' % i, nn.Sequential(*mods))
        for i in range(self._num_outputs):
            if i < self._num_inputs:
                mods = self._level_output()
            else:
                mods = [nn.ReLU(inplace=True)] if self._relu_before_extra else []
                if self._extra_type == 'conv':
                    first_from_input = i == self._num_inputs and self._extra_on_input
                    mods.append(nn.Conv2d(self._num_input_channels_list[-1] if first_from_input else co, co, kernel_size=3,
                                          stride=2, padding=1, bias=True))
                else:
                    mods.append(nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
            self.add_module('fpn_out%d' % i, nn.Sequential(*mods))
        self._init_weights()
        strides = list(self._num_input_strides_list)
        if self._num_outputs <= self._num_inputs:
            strides = strides[:self._num_outputs]
        else:
            # Reference quirk, kept because point_strides are built from this list: fpn.py:100-104 appends
            # `input_strides[-1] * 2**(k+1)` to a list that ALIASES input_strides, so "the last input stride" is the
            # previously appended value from the second extra level on -- [8,16,32] + 2 extras = [8,16,32,64,256], not
            # [..,64,128].  (The reference also grows the caller's list that way; this mirror copies it instead.)
            for k in range(self._num_outputs - self._num_inputs):
                strides.append(strides[-1] * 2 ** (k + 1))
        self._num_output_strides_list = strides

    @property
    def num_output_strides_list(self):
        return self._num_output_strides_list

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                self._init_conv(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    _bottom_up = False

    def forward(self, inputs):
        assert len(inputs) == self._num_inputs
        _lib.require_cuda(inputs[0], '%s.forward' % type(self).__name__)
        if not (torch.is_grad_enabled() and (self.training or any(t.requires_grad for t in inputs))):
            from ... import engine_sibling
            return engine_sibling.neck_forward(self, inputs)
        # training route (autograd over PyTorch-ROCm modules, same parameters)
        lat = [getattr(self, 'lateral%d' % i)(x) for i, x in enumerate(inputs)]
        order = range(self._num_inputs - 1) if self._bottom_up else range(self._num_inputs - 1, 0, -1)
        for i in order:
            dst, src = (i, i + 1) if self._bottom_up else (i - 1, i)
            lat[dst] += nn.functional.interpolate(lat[src], size=lat[dst].shape[2:], mode='nearest')
        outs = []
        for i in range(self._num_outputs):
            if i < self._num_inputs:
                src = lat[i]
            elif i == self._num_inputs and self._extra_on_input:
                src = inputs[-1]
            else:
                src = outs[-1]
            outs.append(getattr(self, 'fpn_out%d' % i)(src))
        return tuple(outs)


class FPN(_PyramidNeck):

    def __init__(self, num_input_channels_list, num_input_strides_list, num_output_channels, num_outputs,
                 extra_on_input=False, extra_type='conv', norm_on_lateral=False, relu_on_lateral=False,
                 relu_before_extra=False, norm_cfg=None):
        super().__init__(num_input_channels_list, num_input_strides_list, num_output_channels, num_outputs, extra_on_input,
                         extra_type, norm_on_lateral, relu_on_lateral, relu_before_extra, norm_cfg)
        self._build()

    def _level_output(self):
        """fpn.py:81-82: every pyramid level is smoothed by a 3x3 conv"""
        c = self._num_output_channels
        return [nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1, bias=True)]

    @staticmethod
    def _init_conv(w):
        nn.init.xavier_uniform_(w, gain=1)        # fpn.py:115-119
