"""LFDResNet -- host-side mirror of the reference backbone's operator interface.

Reference: lfd/model/backbone/lfd_resnet.py (FastBlock :21-93, FasterBlock :96-154,
FastestBlock :157-215, LFDResNet :218-509).  Same constructor kwargs, same parameter
names / shapes / registration order (so `state_dict` round-trips with `strict=True`,
execution/utils.py:53, and `torch.manual_seed(s)` reproduces the reference's initial
weights bit-for-bit), same `num_output_channels_list` / `num_output_strides_list`
(:306-312) and `train()` freeze semantics (:503-509).

The nn.Conv2d / nn.BatchNorm2d children are PARAMETER CONTAINERS: the arithmetic of
`forward` runs in the hand-written gfx950 kernels behind the C-ABI (see ..engine); there
is no PyTorch/CPU compute fallback for inference.
"""
import os

import torch
import torch.nn as nn

__all__ = ['FastBlock', 'FasterBlock', 'FastestBlock', 'LFDResNet']

_NORMS = {'BatchNorm2d': ('num_features', nn.BatchNorm2d), 'GroupNorm': ('num_channels', nn.GroupNorm)}


def build_norm(norm_cfg, channels):
    """norm_cfg is the reference's dict(type='BatchNorm2d'|'GroupNorm', **kwargs)
    (lfd_resnet.py:10-18 evaluates it into nn.<type>(...))."""
    cfg = dict(norm_cfg)
    kind = cfg.pop('type')
    if kind not in _NORMS:
        raise ValueError('unsupported norm type %r' % (kind,))
    key, ctor = _NORMS[kind]
    cfg[key] = channels
    return ctor(**cfg)


def build_activation(activation_cfg):
    cfg = dict(activation_cfg)
    kind = cfg.pop('type')
    return getattr(nn, kind)(**cfg)


def _conv(cin, cout, k, stride, with_norm):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=not with_norm)


class _ResidualBlock(nn.Module):
    """Common shell: registration order `_downsample`, then (_convK, _normK)..., with
    `_activation` registered right after `_norm1` as in the reference blocks."""
    # per-subclass: list of (kernel, in_mult, out_mult) where channel = mult * block channels
    _layout = ()

    def __init__(self, num_input_channels, num_block_channels, stride=1, downsample=None,
                 activation_cfg=dict(type='ReLU', inplace=True), norm_cfg=None):
        super().__init__()
        if downsample is not None:
            assert stride == 2
        if norm_cfg is not None:
            assert norm_cfg['type'] in ['BatchNorm2d', 'GroupNorm']
        self._num_input_channel = num_input_channels
        self._num_block_channel = num_block_channels
        self._stride = stride
        self._activation_cfg = activation_cfg
        self._norm_cfg = norm_cfg
        self._downsample = downsample
        chans = self._channels(num_input_channels, num_block_channels)
        for idx, (k, cin, cout) in enumerate(chans, start=1):
            setattr(self, '_conv%d' % idx, _conv(cin, cout, k, stride if idx == 1 else 1, norm_cfg is not None))
            if norm_cfg is not None:
                setattr(self, '_norm%d' % idx, build_norm(norm_cfg, cout))
            if idx == 1:
                self._activation = build_activation(activation_cfg)
        self.num_convs = len(chans)

    def forward(self, x):
        raise RuntimeError('blocks are parameter containers; run the enclosing LFDResNet')


class FastBlock(_ResidualBlock):
    """3x3 -> 1x1 -> 3x3 (lfd_resnet.py:21-93)."""

    @staticmethod
    def _channels(cin, c):
        return [(3, cin, c), (1, c, c), (3, c, c)]


class FasterBlock(_ResidualBlock):
    """3x3 -> 3x3 (lfd_resnet.py:96-154)."""

    @staticmethod
    def _channels(cin, c):
        return [(3, cin, c), (3, c, c)]


class FastestBlock(_ResidualBlock):
    """3x3 (C/2) -> 3x3 (lfd_resnet.py:157-215)."""

    @staticmethod
    def _channels(cin, c):
        return [(3, cin, c // 2), (3, c // 2, c)]


class LFDResNet(nn.Module):
    mode_to_body_architectures = {'fast': [4, 2, 2, 1, 1], 'faster': [2, 1, 1, 1, 1], 'fastest': [2, 1, 1, 1, 1]}
    mode_to_body_channels = {'fast': [64, 64, 128, 256, 512], 'faster': [64, 64, 128, 128, 256],
                             'fastest': [32, 32, 64, 64, 128]}
    _blocks = {'fast': FastBlock, 'faster': FasterBlock, 'fastest': FastestBlock}

    def __init__(self, block_mode='fast', stem_mode='fast', body_mode='fast', input_channels=3,
                 stem_channels=64, body_architecture=None, body_channels=None,
                 out_indices=((0, 3), (1, 1), (2, 1), (3, 0), (4, 0)), frozen_stages=-1,
                 activation_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='BatchNorm2d'),
                 init_with_weight_file=None, norm_eval=False):
        super().__init__()
        assert block_mode in self._blocks and stem_mode in self._blocks
        assert body_mode in ['fast', 'faster', 'fastest', None]
        if body_mode is None:
            assert body_architecture is not None and body_channels is not None
            arch, chans = body_architecture, body_channels
        else:
            arch = self.mode_to_body_architectures[body_mode]
            chans = self.mode_to_body_channels[body_mode] if body_channels is None else body_channels
        assert len(arch) == len(chans)
        self._block_mode, self._stem_mode = block_mode, stem_mode
        self._input_channels, self._stem_channels = input_channels, stem_channels
        self._out_indices = sorted(out_indices, key=lambda t: (t[0], t[1]))
        for s, b in self._out_indices:
            assert 0 <= s < len(arch) and 0 <= b < arch[s]
        last = max(s for s, _ in self._out_indices)
        self._body_architecture, self._body_channels = list(arch[:last + 1]), list(chans[:last + 1])
        assert frozen_stages <= last + 1
        self._frozen_stages = frozen_stages
        self._activation_cfg, self._norm_cfg = activation_cfg, norm_cfg
        self._init_with_weight_file, self._norm_eval = init_with_weight_file, norm_eval

        self._make_stem()
        self._make_stages()
        self._init_weights()
        if init_with_weight_file is not None:
            assert isinstance(init_with_weight_file, str), 'weight file must be the string path of the file!'
            self._init_with_pretrained_weights()

        stem_stride = 2 if stem_mode == 'fast' else 4
        self._num_output_channels_list = [self._body_channels[s] for s, _ in self._out_indices]
        self._num_output_strides_list = [stem_stride * 2 ** (s + 1) for s, _ in self._out_indices]
        self._engine_owner = None   # set by the enclosing LFD (engine plans span backbone+neck+head)

    # ---- interface used by necks / heads / LFD (lfd_resnet.py:306-312)
    @property
    def num_output_channels_list(self):
        return self._num_output_channels_list

    @property
    def num_output_strides_list(self):
        return self._num_output_strides_list

    # ---- construction
    def stem_spec(self):
        """[(kernel, stride, cin, cout)] of the stem convs (lfd_resnet.py:354-439)."""
        c, i = self._stem_channels, self._input_channels
        return {'fast': [(3, 2, i, c), (1, 1, c, c)],
                'faster': [(3, 2, i, c), (1, 1, c, c), (3, 2, c, c), (1, 1, c, c)],
                'fastest': [(3, 2, i, c // 2), (3, 2, c // 2, c)]}[self._stem_mode]

    def _make_stem(self):
        layers = []
        for k, s, cin, cout in self.stem_spec():
            layers.append(_conv(cin, cout, k, s, self._norm_cfg is not None))
            if self._norm_cfg is not None:
                layers.append(build_norm(self._norm_cfg, cout))
            layers.append(build_activation(self._activation_cfg))
        self._stem = nn.Sequential(*layers)

    def _make_stages(self):
        block = self._blocks[self._block_mode]
        self._block = block
        for i, nblk in enumerate(self._body_architecture):
            c = self._body_channels[i]
            cin = self._stem_channels if i == 0 else self._body_channels[i - 1]
            stage = nn.ModuleList()
            for j in range(nblk):
                if j == 0:
                    ds = [nn.Conv2d(cin, c, kernel_size=1, stride=2, padding=0, bias=self._norm_cfg is None)]
                    if self._norm_cfg is not None:
                        ds.append(build_norm(self._norm_cfg, c))
                    stage.append(block(cin, c, stride=2, downsample=nn.Sequential(*ds),
                                       activation_cfg=self._activation_cfg, norm_cfg=self._norm_cfg))
                else:
                    stage.append(block(c, c, stride=1, downsample=None,
                                       activation_cfg=self._activation_cfg, norm_cfg=self._norm_cfg))
            setattr(self, 'stage%d' % i, stage)

    def _init_weights(self):
        """kaiming_normal_(fan_out, relu) convs, (1, 0) norms -- lfd_resnet.py:342-352."""
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

    def _init_with_pretrained_weights(self):
        """lfd_resnet.py:314-340: strip a leading '*backbone*' key component, non-strict load."""
        path = self._init_with_weight_file
        assert os.path.isfile(path), 'pretrained weight file [{}] does not exist!'.format(path)
        weights = torch.load(path, map_location='cpu')
        renamed = {}
        for k, v in weights['state_dict'].items():
            parts = k.split('.')
            if 'backbone' in parts[0]:
                parts = parts[1:]
            renamed['.'.join(parts)] = v
        missing, unexpected = self.load_state_dict(renamed, strict=False)
        if missing:
            print('[WARNING: ResNet pretrained weights load] missing keys:\n' + '\t'.join(missing))
        if unexpected:
            print('[WARNING: ResNet pretrained weights load] unexpected keys:\n' + '\t'.join(unexpected))

    def _freeze_stages(self):
        if self._frozen_stages > 0:
            self._stem.eval()
            for p in self._stem.parameters():
                p.requires_grad = False
        for i in range(0, self._frozen_stages):
            for blk in getattr(self, 'stage%d' % i):
                blk.eval()
                for p in blk.parameters():
                    p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self._norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    # ---- forward
    def forward(self, x):
        """Returns the tuple of tapped NCHW fp32 maps (lfd_resnet.py:488-501), computed by the
        HIP engine.  Stand-alone use builds a backbone-only plan."""
        from ...engine import backbone_only_forward
        return backbone_only_forward(self, x)
