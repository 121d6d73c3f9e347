"""Layer-by-layer gfx950 engine for the sibling meta-architectures: FCOS (lfd/model/fcos.py:414-449) and
LFDv2 (lfd/model/lfdv2.py:671-702) over an LFDResNet backbone, an FPN / SimpleFPN / SimpleNeck neck
(lfd/model/neck/fpn.py:127-152, simple_fpn.py:141-172, simple_neck.py) and an FCOSHead / LFDHead with 1x1 or 3x3 towers
(lfd/model/head/fcos_head.py:129-154, lfd_head.py:164-185).

The LFD configurations the reference ships run on the fused plan of engine.py (stem / residual-block / 3-pass head
kernels).  The siblings are not used by any shipped config; they get the same kernels one layer at a time:

  backbone             engine.EnginePlan.run_backbone (the fused stem / block / conv kernels, unchanged)
  conv (+ folded BN)   lfd_conv2d_nhwc_f16                     (MFMA implicit GEMM, csrc/conv_impl.h)
  GroupNorm (+ ReLU)   lfd_gn_train_stats_f16 + lfd_gn_train_apply_f16   (GroupNorm has no eval mode: same kernels as training)
  top-down merge       lfd_upsample_nearest_add_nhwc_f16       (csrc/sibling.hip)
  extra levels         lfd_relu_inplace_f16, conv 3x3 s2 | lfd_maxpool3x3s2_nhwc_f16
  output convs         lfd_conv2d_nhwc_f16_acc32 (fp32 logits) + lfd_pack_level_outputs_f32 (Scale, exp, [N,P,C] concat)

Inter-layer storage is NHWC fp16 like the LFD plan.  Widths: 32 / 64 / 128 channels (others zero-padded up, > 128
rejected); GroupNorm with 8 channels per group (the kernels' group size: 128/16, 64/8).  No PyTorch fallback: an
unsupported module raises.
"""
import torch
import torch.nn as nn

from . import _lib, engine, ops

_UNION = ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss')


def _unsupported(msg):
    raise RuntimeError('lfd_amd sibling engine: unsupported configuration: %s (no PyTorch fallback for inference)' % msg)


class _Conv(object):
    """conv [+ eval BatchNorm folded] [+ GroupNorm] [+ ReLU] on an NHWC fp16 map whose channel count is `cin_have`
    (>= conv.in_channels: the producer's zero padding)."""

    def __init__(self, conv, norm, relu, cin_have, dev, f32out=False):
        if not isinstance(conv, nn.Conv2d):
            _unsupported('expected nn.Conv2d, got %s' % type(conv).__name__)
        ks, st = conv.kernel_size[0], conv.stride[0]
        if conv.kernel_size != (ks, ks) or conv.stride != (st, st) or conv.padding != (ks // 2, ks // 2) \
                or ks not in (1, 3) or st not in (1, 2) or conv.groups != 1 or conv.dilation != (1, 1):
            _unsupported('conv %s' % (conv,))
        bn = norm if isinstance(norm, nn.BatchNorm2d) else None
        self.gn = norm if isinstance(norm, nn.GroupNorm) else None
        if norm is not None and bn is None and self.gn is None:
            _unsupported('norm %s' % type(norm).__name__)
        w, b = engine._fold_conv_norm(conv, bn)
        self.cout_real = w.shape[0]
        cout = engine.pad_channels(max(self.cout_real, 32))        # output convs have 1 .. 6 filters: one 32-wide MFMA tile
        if f32out and cin_have == 128 and cout == 32:
            cout = 64                     # the 128-channel fp32-output kernels are instantiated for >= 64 outputs
        if self.gn is not None:
            if cout != self.cout_real or self.cout_real != 8 * self.gn.num_groups:
                _unsupported('GroupNorm(%d, %d): the kernels take 8 channels per group on 32/64/128 channels'
                             % (self.gn.num_groups, self.cout_real))
            self.gamma = (self.gn.weight.detach().float() if self.gn.weight is not None
                          else torch.ones(cout)).to(dev).contiguous()
            self.beta = (self.gn.bias.detach().float() if self.gn.bias is not None
                         else torch.zeros(cout)).to(dev).contiguous()
        wp = torch.zeros((cout, cin_have, ks, ks), dtype=torch.float32, device=w.device)
        wp[:w.shape[0], :w.shape[1]] = w
        bp = torch.zeros(cout, dtype=torch.float32, device=w.device)
        bp[:b.shape[0]] = b
        self.w = ops.pack_conv_weight(wp).to(dev)
        self.b = bp.to(dev).contiguous()
        self.cin, self.cout, self.ks, self.stride, self.relu, self.f32out = cin_have, cout, ks, st, bool(relu), f32out

    def __call__(self, x):
        if self.f32out:
            return ops.conv2d_nhwc_f32out(x, self.w, self.b, self.cin, self.cout, self.ks, self.stride)
        y = ops.conv2d_nhwc(x, self.w, self.b, self.cin, self.cout, self.ks, self.stride, self.relu and self.gn is None)
        if self.gn is not None:
            stats = ops.gn_train_stats(y, self.gn.num_groups, self.gn.eps)
            y = ops.gn_train_apply(y, self.gn.num_groups, stats, self.gamma, self.beta, relu=self.relu)
        return y


def _split_sequential(seq):
    """nn.Sequential of [conv, (norm), (ReLU)] groups / bare ReLU / MaxPool -> list of ('conv', conv, norm, relu) |
    ('relu',) | ('pool',) steps"""
    mods = list(seq)
    steps, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            norm, relu, j = None, False, i + 1
            if j < len(mods) and isinstance(mods[j], (nn.BatchNorm2d, nn.GroupNorm)):
                norm, j = mods[j], j + 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu, j = True, j + 1
            steps.append(('conv', m, norm, relu))
            i = j
        elif isinstance(m, nn.ReLU):
            steps.append(('relu',))
            i += 1
        elif isinstance(m, nn.MaxPool2d):
            if (m.kernel_size, m.stride, m.padding) != (3, 2, 1):
                _unsupported('MaxPool2d %s' % (m,))
            steps.append(('pool',))
            i += 1
        else:
            _unsupported('module %s in a neck / head path' % type(m).__name__)
    return steps


class _Path(object):
    """a compiled nn.Sequential"""

    def __init__(self, seq, cin_have, dev, f32out_last=False):
        self.ops = []
        c = cin_have
        steps = _split_sequential(seq)
        for k, s in enumerate(steps):
            if s[0] == 'conv':
                layer = _Conv(s[1], s[2], s[3], c, dev, f32out=f32out_last and k == len(steps) - 1)
                self.ops.append(layer)
                c = layer.cout
            else:
                self.ops.append(s[0])
        self.cout = c
        self.inplace_relu_first = bool(steps) and steps[0][0] == 'relu'

    def __call__(self, x):
        for o in self.ops:
            if o == 'relu':
                x = ops.relu_(x)          # nn.ReLU(inplace=True): the caller's tensor changes too (by design, see plan)
            elif o == 'pool':
                x = ops.maxpool3x3s2(x)
            else:
                x = o(x)
        return x


class SiblingPlan(object):
    """Compiled (neck, head) of a sibling model for one device; the backbone keeps its own engine.EnginePlan."""

    def __init__(self, backbone, neck, head, dev):
        self.param_sig = engine._param_signature(backbone, neck, head)
        self.dev = dev
        self.bb_plan = engine.get_plan(backbone, backbone, None, None, dev)
        self.tap_real = list(self.bb_plan.tap_channels)
        self.tap_have = [engine.pad_channels(c) for c in self.tap_real]      # the plan's buffers carry padded channels
        self._build_neck(neck)
        self._build_head(head)

    # ------------------------------------------------------------------ neck
    def _build_neck(self, neck):
        kind = type(neck).__name__
        self.neck_kind = kind
        if kind == 'SimpleNeck':
            self.laterals = [_Path(getattr(neck, 'neck%d' % i), c, self.dev) for i, c in enumerate(self.tap_have)]
            self.outs, self.num_inputs, self.num_outputs = None, len(self.laterals), len(self.laterals)
            self.bottom_up = False
            self.merge = False
            return
        if kind not in ('FPN', 'SimpleFPN'):
            _unsupported('neck %s' % kind)
        self.merge = True
        self.num_inputs, self.num_outputs = neck._num_inputs, neck._num_outputs
        self.extra_on_input = neck._extra_on_input
        self.bottom_up = bool(getattr(neck, '_neighbouring_mode', False))
        self.laterals = [_Path(getattr(neck, 'lateral%d' % i), c, self.dev) for i, c in enumerate(self.tap_have)]
        cn = self.laterals[0].cout
        self.outs, self.out_real = [], []
        for i in range(self.num_outputs):
            from_input = i == self.num_inputs and self.extra_on_input
            src_c = self.tap_have[-1] if from_input else (self.outs[-1].cout if i >= self.num_inputs else cn)
            path = _Path(getattr(neck, 'fpn_out%d' % i), src_c, self.dev)
            self.outs.append(path)
            if any(isinstance(o, _Conv) for o in path.ops) or i < self.num_inputs:
                self.out_real.append(neck._num_output_channels)
            else:       # a pooled extra level keeps the channels of what it pools (the last input, or the previous level)
                self.out_real.append(neck._num_input_channels_list[-1] if from_input else self.out_real[-1])

    def run_neck(self, taps):
        """taps: NHWC fp16 maps of the backbone.  Mirrors FPN.forward / SimpleFPN.forward including their in-place
        semantics: `+=` on the lateral maps and the inplace ReLU in front of an extra level (which also rewrites the
        level it reads from -- fpn.py:66-79 builds Sequential(ReLU(inplace=True), conv) and feeds fpn_outputs[-1])."""
        lat = [p(t) for p, t in zip(self.laterals, taps)]
        if not self.merge:
            return lat
        if self.bottom_up:
            for i in range(self.num_inputs - 1):
                ops.upsample_nearest_add_(lat[i], lat[i + 1])
        else:
            for i in range(self.num_inputs - 1, 0, -1):
                ops.upsample_nearest_add_(lat[i - 1], lat[i])
        outs = []
        for i in range(self.num_outputs):
            if i == self.num_inputs:
                src = taps[-1] if self.extra_on_input else outs[-1]
            elif i > self.num_inputs:
                src = outs[-1]
            else:
                src = lat[i]
            outs.append(self.outs[i](src))
        return outs

    # ------------------------------------------------------------------ head
    def _build_head(self, head):
        kind = type(head).__name__
        self.head_kind = kind
        cn = self.laterals[0].cout
        self.levels = []
        if kind == 'FCOSHead':
            cls_t = _Path(nn.Sequential(*list(head._classification_path)), cn, self.dev)
            reg_t = _Path(nn.Sequential(*list(head._regression_path)), cn, self.dev)
            cls_o = _Conv(head._classification, None, False, cls_t.cout, self.dev, f32out=True)
            ctr_o = _Conv(head._centerness, None, False, cls_t.cout, self.dev, f32out=True)
            reg_o = _Conv(head._regression, None, False, reg_t.cout, self.dev, f32out=True)
            self.num_cls_channels = head._num_classes
            for i in range(head._num_heads):
                self.levels.append(dict(cls_t=cls_t, reg_t=reg_t, cls_o=cls_o, ctr_o=ctr_o, reg_o=reg_o,
                                        scale=float(head._scales[i]._scale.detach()), exp=True))
            self.has_ctr = True
            return
        if kind not in ('LFDHead', 'LFDHeadV1'):
            _unsupported('head %s' % kind)
        self.has_ctr = False
        self.num_cls_channels = head.num_cls_channels
        union = head._regression_loss_type in _UNION
        cache = {}

        def compiled(seq, cin, f32out_last=False):
            key = (id(seq), cin, f32out_last)
            if key not in cache:
                cache[key] = _Path(seq, cin, self.dev, f32out_last=f32out_last)
            return cache[key]

        for i in range(head._num_heads):
            merge = getattr(head, 'head%d_merge_path' % i)
            cls_p = getattr(head, 'head%d_classification_path' % i)
            reg_p = getattr(head, 'head%d_regression_path' % i)
            # (a parameter update changes the plan signature, so the value read here cannot go stale)
            lv = dict(exp=False, scale=float(head._scales[i]._scale.detach()) if union else 1.0, ctr_o=None)
            if head._merge_path_flag:
                lv['merge'] = compiled(merge, cn)
                c = lv['merge'].cout
            else:
                lv['merge'] = None
                c = cn
            if kind == 'LFDHeadV1':     # towers end without the output convs: those are per-level members
                lv['cls_p'], lv['reg_p'] = compiled(cls_p, c), compiled(reg_p, c)
                lv['cls_o'] = _Conv(head._classifiers[i], None, False, lv['cls_p'].cout, self.dev, f32out=True)
                lv['reg_o'] = _Conv(head._regressors[i], None, False, lv['reg_p'].cout, self.dev, f32out=True)
            else:
                lv['cls_p'] = compiled(cls_p, c, True)
                lv['reg_p'] = compiled(reg_p, c, True)
            self.levels.append(lv)

    def run_head(self, feats):
        """-> (cls [N,P,C'] fp32, reg [N,P,4] fp32, centerness [N,P,1] fp32 | None, sizes)"""
        n = feats[0].shape[0]
        sizes = [(f.shape[1], f.shape[2]) for f in feats]
        P = sum(h * w for h, w in sizes)
        dev = feats[0].device
        cls = torch.empty((n, P, self.num_cls_channels), dtype=torch.float32, device=dev)
        reg = torch.empty((n, P, 4), dtype=torch.float32, device=dev)
        ctr = torch.empty((n, P, 1), dtype=torch.float32, device=dev) if self.has_ctr else None
        p0 = 0
        for lv, f, (h, w) in zip(self.levels, feats, sizes):
            scale = lv['scale']
            if self.head_kind == 'FCOSHead':
                tc = lv['cls_t'](f)
                tr = lv['reg_t'](f)
                ops.pack_level_outputs(lv['cls_o'](tc), cls, 0, self.num_cls_channels, p0)
                ops.pack_level_outputs(lv['ctr_o'](tc), ctr, 0, 1, p0)
                ops.pack_level_outputs(lv['reg_o'](tr), reg, 0, 4, p0, scale=scale, exp=True)
            else:
                t = lv['merge'](f) if lv['merge'] is not None else f
                c, r = lv['cls_p'](t), lv['reg_p'](t)
                if 'cls_o' in lv:
                    c, r = lv['cls_o'](c), lv['reg_o'](r)
                ops.pack_level_outputs(c, cls, 0, self.num_cls_channels, p0)
                ops.pack_level_outputs(r, reg, 0, 4, p0, scale=scale)
            p0 += h * w
        return cls, reg, ctr, sizes


def get_plan(owner, backbone, neck, head, device):
    cache = owner.__dict__.setdefault('_lfd_sibling_cache', {})
    key = (device.type, device.index)
    plan = cache.get(key)
    sig = engine._param_signature(backbone, neck, head)
    if plan is None or plan.param_sig != sig:
        with torch.no_grad():
            plan = SiblingPlan(backbone, neck, head, device)
        cache[key] = plan
    return plan


def backbone_taps(plan, x):
    """image batch -> the backbone's tapped maps, NHWC fp16 (engine-owned buffers, padded channels)"""
    fmt, n, h, w = engine._input_format(x)
    st = plan.bb_plan.state_for(n, h, w)
    plan.bb_plan.run_backbone(x, fmt, st)
    return [st.bufs[t] for t in plan.bb_plan.taps]


@torch.no_grad()
def sibling_forward(model, x, use_graph=False):
    """eval-mode forward of FCOS / LFDv2 on the device -> (cls, reg, centerness | None, sizes), fp32, level-concatenated.
    use_graph: the ~100 launches of a forward are captured into one HIP graph per input buffer and replayed (the
    returned tensors are then graph-owned and overwritten by the next call with the same input buffer)."""
    _lib.require_cuda(x, '%s.forward' % type(model).__name__)
    if not x.is_contiguous():
        x = x.contiguous()
    plan = get_plan(model, model._backbone, model._neck, model._head, x.device)

    def run():
        return plan.run_head(plan.run_neck(backbone_taps(plan, x)))

    with torch.cuda.device(x.device):
        if not use_graph:
            return run()
        graphs = plan.__dict__.setdefault('graphs', {})
        key = (x.data_ptr(), tuple(x.shape), x.dtype)
        ent = graphs.get(key)
        if ent is None:
            if len(graphs) >= 4:
                graphs.pop(next(iter(graphs)))
            run()                               # warm-up outside capture: kernel attributes, workspaces
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = run()
            ent = (g, outs, x)                  # keep the captured input alive
            graphs[key] = ent
        ent[0].replay()
        return ent[1]


@torch.no_grad()
def neck_forward(neck, inputs):
    """FPN / SimpleFPN .forward stand-alone on NCHW fp32 feature maps (reference types in and out, fpn.py:127-152); the
    layout / precision conversion around the kernels is plain tensor plumbing."""
    x0 = inputs[0]
    _lib.require_cuda(x0, '%s.forward' % type(neck).__name__)
    dev = x0.device
    cache = neck.__dict__.setdefault('_lfd_sibling_cache', {})
    sig = engine._param_signature(neck)
    key = (dev.type, dev.index)
    ent = cache.get(key)
    if ent is None or ent.param_sig != sig:
        ent = SiblingPlan.__new__(SiblingPlan)
        ent.param_sig, ent.dev = sig, dev
        ent.tap_have = [engine.pad_channels(t.shape[1]) for t in inputs]
        ent._build_neck(neck)
        cache[key] = ent
    with torch.cuda.device(dev):
        taps = []
        for t, c in zip(inputs, ent.tap_have):
            y = t.permute(0, 2, 3, 1).half()
            if c != t.shape[1]:
                y = torch.nn.functional.pad(y, (0, c - t.shape[1]))
            taps.append(y.contiguous())
        outs = ent.run_neck(taps)
    real = ent.out_real if ent.merge else [neck._num_neck_channels] * len(outs)
    return tuple(o[..., :c].permute(0, 3, 1, 2).float() for o, c in zip(outs, real))
