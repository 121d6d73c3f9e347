"""SimpleFPN -- host-side mirror of lfd/model/neck/simple_fpn.py:22-172: FPN without the 3x3 smoothing convs (a pyramid
level's output IS its merged lateral map), optional bottom-up `neighbouring_mode` (lateral[i] += upsample(lateral[i+1])
walking from the finest level, :147-151), kaiming-normal init (:129-139).  Execution: see fpn.py."""
import torch.nn as nn

from .fpn import _PyramidNeck

__all__ = ['SimpleFPN']


class SimpleFPN(_PyramidNeck):

    def __init__(self, num_input_channels_list, num_input_strides_list, num_output_channels, num_outputs,
                 extra_on_input=False, extra_type='conv', norm_on_lateral=False, relu_on_lateral=False,
                 relu_before_extra=True, norm_cfg=None, neighbouring_mode=False):
        super().__init__(num_input_channels_list, num_input_strides_list, num_output_channels, num_outputs, extra_on_input,
                         extra_type, norm_on_lateral, relu_on_lateral, relu_before_extra, norm_cfg)
        self._neighbouring_mode = neighbouring_mode
        if neighbouring_mode:
            assert num_outputs + 1 >= self._num_inputs
        self._bottom_up = bool(neighbouring_mode)
        self._build()

    def _level_output(self):
        return []                                  # simple_fpn.py:100-101: "do nothing"

    @staticmethod
    def _init_conv(w):
        nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')
