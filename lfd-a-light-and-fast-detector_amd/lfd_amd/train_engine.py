"""Training-mode forward / backward of the LFDResNet backbone on the hand-written gfx950 kernels (SURVEY 8a
row 18; csrc/train.hip + the MFMA conv of csrc/conv.hip), bridged into autograd as ONE node so that the torch
modules after it (neck / head, not yet hand-written in training mode) and the loss drive it.

Reference semantics: LFDResNet.forward in train mode (lfd/model/backbone/lfd_resnet.py:488-501) over
nn.Conv2d(bias=False) -> nn.BatchNorm2d (batch statistics, running statistics updated) -> ReLU units, residual
blocks `relu(norm(conv(..)) + identity)` (:96-154), identity = norm(conv1x1 s2(x)) for the first block of a stage
(:458-468).  Numerics here: NHWC fp16 activations / activation gradients (gradients carry a power-of-two loss
scale), fp32 accumulation on MFMA, fp32 statistics, parameters and parameter gradients.

The backward is a hand-written schedule over the recorded units (no autograd inside):
    g, dy = bn_bwd(dz)  ->  dW = wgrad(x, dy)  ->  dx = conv(dy, W^T flipped) (+ gradient already collected for x)
"""
import os

import torch
import torch.nn as nn

from . import ops

LOSS_SCALE = 1024.0   # initial value; power of two; activation gradients are stored as fp16 * loss_scale()
_scale_state = {'value': LOSS_SCALE}


def loss_scale():
    """the loss scale the next backward pass multiplies the prediction gradients with (fp16 activation gradients carry it,
    the weight-gradient / norm kernels divide it out again: parameter gradients are unscaled fp32)"""
    return _scale_state['value']


def set_loss_scale(v):
    """power of two in [1, 65536] (lfd_amd.train.DynamicLossScale drives this from the gradient norm's finiteness)"""
    v = float(v)
    if not (1.0 <= v <= 65536.0) or v != 2.0 ** round(__import__('math').log2(v)):
        raise ValueError('loss scale must be a power of two in [1, 65536]')
    _scale_state['value'] = v


class _Unit(object):
    __slots__ = ('conv', 'norm', 'relu', 'src', 'res', 'dst', 'first', 'level')


class _Out(object):
    """Output convs of one pyramid level: cls conv1x1 (+bias) and reg conv1x1 (+bias) x Scale (lfd_head.py:107-111,
    :176-180).  With a merged tower both read the same activation and run as ONE conv (rows concatenated, padded to 32)."""
    __slots__ = ('level', 'convs', 'src', 'scale')


def _conv_ok(m):
    return m.in_channels == 3 or (m.in_channels in (32, 64, 128) and m.out_channels in (32, 64, 128))


def supported(backbone):
    """The HIP training path covers BatchNorm2d + ReLU backbones with 32/64/128-channel convs, nothing frozen."""
    if backbone._norm_cfg is None or backbone._norm_cfg.get('type') != 'BatchNorm2d':
        return False
    if backbone._activation_cfg.get('type') != 'ReLU' or backbone._frozen_stages > 0 or backbone._norm_eval:
        return False
    if backbone._input_channels != 3:
        return False
    # the hand-written backward accumulates into EVERY covered parameter's .grad: a frozen parameter (requires_grad=False,
    # however it was frozen) sends the module through PyTorch-ROCm autograd instead
    if not all(p.requires_grad for p in backbone.parameters()):
        return False
    for m in backbone.modules():
        if isinstance(m, nn.Conv2d) and not _conv_ok(m):
            return False
        if isinstance(m, nn.BatchNorm2d) and (m.momentum is None or not m.affine or not m.track_running_stats
                                              or not m.training):
            return False          # incl. norm layers switched to eval() inside a training model: batch statistics only here
    first = backbone._stem[0]
    return first.out_channels in (32, 64) and first.kernel_size == (3, 3) and first.stride == (2, 2)


def network_supported(model):
    """+ SimpleNeck with BatchNorm2d, LFDHead with 1x1 convs and GroupNorm groups of 8 channels, <= 60 output channels."""
    bb, neck, head = model._backbone, model._neck, model._head
    if type(neck).__name__ != 'SimpleNeck' or type(head).__name__ != 'LFDHead':
        return False          # FPN / SimpleFPN necks, LFDHeadV1, FCOSHead: PyTorch-ROCm autograd behind the HIP backbone
    if not supported(bb):
        return False
    if not all(p.requires_grad for p in list(neck.parameters()) + list(head.parameters())):
        return False
    if neck._norm_cfg is None or neck._norm_cfg.get('type') != 'BatchNorm2d' or neck._activation_cfg.get('type') != 'ReLU':
        return False
    if head._norm_cfg is None or head._norm_cfg.get('type') != 'GroupNorm' or head._activation_cfg.get('type') != 'ReLU':
        return False
    if head._conv_kernel_size != 1 or head._num_head_channels != 8 * head._norm_cfg.get('num_groups', 0):
        return False
    if neck._num_neck_channels not in (64, 128) or head._num_head_channels not in (64, 128):
        return False
    for m in head.modules():
        if isinstance(m, nn.GroupNorm) and not m.affine:
            return False
    for m in neck.modules():
        if isinstance(m, nn.BatchNorm2d) and (m.momentum is None or not m.affine or not m.track_running_stats
                                              or not m.training):
            return False
        if isinstance(m, nn.Conv2d) and not _conv_ok(m):
            return False
    return head.num_cls_channels + 4 <= 64


class _Builder(object):
    def __init__(self):
        self.units, self.n_act, self.level = [], 1, None

    def add(self, conv, norm, relu, src, res=None):
        u = _Unit()
        u.conv, u.norm, u.relu, u.src, u.res, u.dst, u.first = conv, norm, relu, src, res, self.n_act, src == 0
        u.level = self.level            # None: backbone; i: the neck / tower chain of pyramid level i
        self.units.append(u)
        self.n_act += 1
        return u.dst


def _build_backbone(b, backbone):
    cur = 0
    mods = list(backbone._stem)
    for i in range(0, len(mods), 3):          # (conv, norm, activation) triples
        cur = b.add(mods[i], mods[i + 1], True, cur)
    taps = {}
    want = [tuple(t) for t in backbone._out_indices]
    for si, nblk in enumerate(backbone._body_architecture):
        for bi in range(nblk):
            blk = getattr(backbone, 'stage%d' % si)[bi]
            ident = cur
            if blk._downsample is not None:
                ident = b.add(blk._downsample[0], blk._downsample[1], False, cur)
            a = cur
            for ci in range(1, blk.num_convs + 1):
                last = ci == blk.num_convs
                a = b.add(getattr(blk, '_conv%d' % ci), getattr(blk, '_norm%d' % ci), True, a, ident if last else None)
            cur = a
            if (si, bi) in want:
                taps[(si, bi)] = cur
    return [taps[t] for t in want]


def build_units(backbone):
    """-> (units, tap activation indices).  Activation 0 is the input image."""
    b = _Builder()
    taps = _build_backbone(b, backbone)
    return b.units, taps


def build_network(model):
    """-> (units, outs): backbone + neck + head towers as conv/norm/ReLU units, and the per-level output convs."""
    b = _Builder()
    taps = _build_backbone(b, model._backbone)
    neck, head = model._neck, model._head
    outs = []

    def tower(path, cur):
        mods = list(path)
        n3 = len(mods) - (1 if mods and isinstance(mods[-1], nn.Conv2d) else 0)
        for i in range(0, n3, 3):
            cur = b.add(mods[i], mods[i + 1], True, cur)
        return cur

    for i, tap in enumerate(taps):
        b.level = i
        nk = getattr(neck, 'neck%d' % i)
        cur = b.add(nk[0], nk[1], True, tap)
        cur = tower(getattr(head, 'head%d_merge_path' % i), cur)
        cpath, rpath = getattr(head, 'head%d_classification_path' % i), getattr(head, 'head%d_regression_path' % i)
        csrc, rsrc = tower(cpath, cur), tower(rpath, cur)
        scale = head._scales[i] if hasattr(head, '_scales') else None
        if csrc == rsrc:
            o = _Out()
            o.level, o.convs, o.src, o.scale = i, [('cls', cpath[-1]), ('reg', rpath[-1])], csrc, scale
            outs.append(o)
        else:
            for kind, conv, src in (('cls', cpath[-1], csrc), ('reg', rpath[-1], rsrc)):
                o = _Out()
                o.level, o.convs, o.src, o.scale = i, [(kind, conv)], src, scale
                outs.append(o)
    return b.units, outs


def _unique(params):
    seen, out = set(), []
    for p in params:
        if id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


def backbone_params(units):
    ps = []
    for u in units:
        ps += [u.conv.weight, u.norm.weight, u.norm.bias]
    return _unique(ps)


def network_params(units, outs):
    ps = backbone_params(units)
    for o in outs:
        for _, conv in o.convs:
            ps += [conv.weight, conv.bias]
        if o.scale is not None:
            ps.append(o.scale._scale)
    return _unique(ps)


def _dgrad_weight(w):
    """weights of the conv that maps dL/dy to dL/dx: swap the channel roles, flip the taps"""
    return ops.pack_conv_weight_train(w, data_gradient=True)


_pack_batches = {}


class _Packs(object):
    """fp16 fragment packs of the conv weights of a pass.  With `units` given, every distinct conv weight of the schedule
    is packed by ONE launch up front (ops.PackBatch, job table cached per schedule); other weights (the concatenated
    output convs) are packed individually on first use.  Shared by the pyramid levels within a pass."""

    def __init__(self, units=None, data_gradient=False):
        self.c = {}
        if units:
            ws, seen = [], set()
            for u in units:
                w = u.conv.weight
                if not u.first and id(w) not in seen:
                    seen.add(id(w))
                    ws.append(w.detach())
            if ws:
                key = (id(units), data_gradient)
                pb = _pack_batches.get(key)
                if pb is None or not pb.matches(ws):
                    if len(_pack_batches) > 32:
                        _pack_batches.clear()
                    pb = ops.PackBatch(ws, data_gradient)
                    _pack_batches[key] = pb
                for w, out in zip(ws, pb.run()):
                    self.c[(w.data_ptr(), data_gradient)] = out

    def __call__(self, w, data_gradient=False):
        k = (w.data_ptr(), data_gradient)
        if k not in self.c:
            self.c[k] = ops.pack_conv_weight_train(w, data_gradient=data_gradient)
        return self.c[k]


_zero_cache = {}


class _Zeros(object):
    """zero bias vectors (the units' convs have none): persistent per device -- nobody writes them"""

    def __init__(self, dev):
        self.dev = dev

    def __call__(self, n):
        k = (self.dev.type, self.dev.index, n)
        if k not in _zero_cache:
            _zero_cache[k] = torch.zeros(n, dtype=torch.float32, device=self.dev)
        return _zero_cache[k]


def forward(units, tap_ids, x):
    """x: NCHW fp32 image batch (as LFD.forward receives it, lfd.py:511).  -> (requested activations NHWC fp16, saved)."""
    acts = {0: x}
    tape = []
    zeros = _Zeros(x.device)
    packs = _Packs(units, False)
    fused_stats = os.environ.get('LFD_CONV_BN_STATS', '1') == '1'     # A/B switch: 0 = conv, then a statistics pass over y
    for u in units:
        conv, norm = u.conv, u.norm
        xin = acts[u.src]
        ks, st = conv.kernel_size[0], conv.stride[0]
        stats = None
        if u.first and isinstance(norm, nn.BatchNorm2d) and fused_stats:
            y, stats = ops.stem_conv0_train_fwd_bn_stats(xin, conv.weight, norm.eps, norm.momentum, norm.running_mean,
                                                         norm.running_var)
        elif u.first:
            y = ops.stem_conv0_train_fwd(xin, conv.weight)
        elif isinstance(norm, nn.BatchNorm2d) and fused_stats:
            # the batch statistics come out of the conv's epilogue: no separate read of y (csrc/conv_stats.hip)
            cout = conv.out_channels
            y, stats = ops.conv2d_bn_stats(xin, packs(conv.weight), zeros(cout), conv.in_channels, cout, ks, st, norm.eps,
                                           norm.momentum, norm.running_mean, norm.running_var)
        else:
            cout = conv.out_channels
            y = ops.conv2d_nhwc(xin, packs(conv.weight), zeros(cout), conv.in_channels, cout, ks, st, False)
        if isinstance(norm, nn.GroupNorm):
            stats = ops.gn_train_stats(y, norm.num_groups, norm.eps)
            z = ops.gn_train_apply(y, norm.num_groups, stats, norm.weight.detach(), norm.bias.detach(), u.relu)
        else:
            if stats is None:
                stats = ops.bn_train_stats(y, norm.eps, norm.momentum, norm.running_mean, norm.running_var)
            z = ops.bn_train_apply(y, stats, norm.weight.detach(), norm.bias.detach(),
                                   acts[u.res] if u.res is not None else None, u.relu)
        acts[u.dst] = z
        tape.append((y, stats))
    torch._foreach_add_([u.norm.num_batches_tracked for u in units if isinstance(u.norm, nn.BatchNorm2d)], 1)
    return [acts[t] for t in tap_ids], (acts, tape)


class _GradStore(object):
    """Where the fp32 parameter gradients go.  in_place: straight into `p.grad` (created zeroed when missing) -- the
    kernels `+=` into it, which is autograd's AccumulateGrad semantics without one temporary + one add per tensor (the
    autograd node then reports no gradient for the parameters).  Otherwise: private zero-initialised buffers by
    parameter identity (tests).  Shared modules (the head towers of all levels) accumulate either way."""

    def __init__(self, in_place=False):
        self.g, self.in_place = {}, in_place

    def target(self, p):
        """the buffer every kernel accumulates this parameter's gradient into"""
        if self.in_place:
            if p.grad is None:
                p.grad = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
            return p.grad
        k = id(p)
        if k not in self.g:
            self.g[k] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        return self.g[k]

    def add(self, p, grad):
        self.target(p).add_(grad.reshape(p.shape))

    def get(self, p):
        return p.grad if self.in_place else self.g.get(id(p))


def backward(units, saved, grads, scale=None, store=None, trace=None):
    """grads: {activation index: dL/dact NHWC fp16 multiplied by `scale`} for the activations consumed outside the units
    (taps for the backbone alone, tower outputs for the whole network).  -> _GradStore of fp32 parameter gradients.
    trace: optional list that receives the per-unit tensors (tests check every unit against PyTorch given the same
    inputs)."""
    acts, tape = saved
    scale = loss_scale() if scale is None else scale
    inv = 1.0 / scale
    grads = dict(grads)
    store = store if store is not None else _GradStore()
    zeros = _Zeros(acts[0].device)
    packs = _Packs(units, True)
    for ui in range(len(units) - 1, -1, -1):
        u = units[ui]
        dz = grads.pop(u.dst, None)
        if dz is None:
            continue
        conv, norm = u.conv, u.norm
        y, stats = tape[ui]
        z = acts[u.dst] if u.relu else None
        g = None
        dgamma, dbeta, dw = store.target(norm.weight), store.target(norm.bias), store.target(conv.weight)
        snap = [t.clone() for t in (dgamma, dbeta, dw)] if trace is not None else None
        if isinstance(norm, nn.GroupNorm):
            dy = ops.gn_train_backward(dz, y, z, norm.num_groups, stats, norm.weight.detach(), inv, dgamma, dbeta, True)
        else:
            # without a residual input the ReLU mask is recomputed from y (one tensor less to read in both passes)
            dy, g = ops.bn_train_backward(dz, y, z if u.res is not None else None, stats, norm.weight.detach(), inv, dgamma,
                                          dbeta, want_g=u.res is not None, accumulate=True, relu=u.relu,
                                          beta=norm.bias.detach())
        if u.res is not None:
            grads[u.res] = g if u.res not in grads else grads[u.res] + g
        xin = acts[u.src]
        ks, st = conv.kernel_size[0], conv.stride[0]
        rec = dict(ui=ui, dz=dz, dy=dy, g=g, dx_prev=grads.get(u.src), dx=None) if trace is not None else None
        if u.first:
            ops.stem_conv0_wgrad(xin, dy, inv, out=dw, accumulate=True)
        else:
            ops.conv_wgrad(xin, dy, ks, st, inv, out=dw, accumulate=True)
            cin = conv.in_channels
            if (st == 2 and ks == 3 and cin == 64 and conv.out_channels == 64 and os.environ.get('LFD_DGRAD_S2', '1') == '1'):
                # per output parity, 9 tap-products per 2 x 2 pixels instead of 36 and no zero-inserted tensor (csrc/dgrad_s2.hip)
                grads[u.src] = ops.conv3x3s2_dgrad(dy, packs(conv.weight, True), xin.size(1), xin.size(2), residual=grads.get(u.src))
            else:
                if st == 2:
                    dy = ops.zero_insert2(dy, xin.size(1), xin.size(2))
                grads[u.src] = ops.conv2d_nhwc(dy, packs(conv.weight, True), zeros(cin), conv.out_channels, cin, ks, 1,
                                               False, residual=grads.get(u.src))
            if rec is not None:
                rec['dx'] = grads[u.src]
        if rec is not None:      # this unit's own contribution (the buffers accumulate over shared modules)
            rec.update(dgamma=dgamma - snap[0], dbeta=dbeta - snap[1], dw=dw - snap[2])
            trace.append(rec)
    return store


# ---------------------------------------------------------------------------------------------- output convs
def _out_weight(o):
    """rows of the level's output convs concatenated and zero-padded to 64 (the narrowest 1x1 conv shape instantiated
    for 128 input channels, forward and data gradient): [64, C, 1, 1], bias [64]"""
    w = torch.cat([c.weight.detach() for _, c in o.convs], 0)
    bias = torch.cat([c.bias.detach() for _, c in o.convs], 0)
    rows = -(-w.size(0) // 64) * 64
    wp = torch.zeros((rows,) + tuple(w.shape[1:]), dtype=torch.float32, device=w.device)
    wp[:w.size(0)] = w
    bp = torch.zeros(rows, dtype=torch.float32, device=w.device)
    bp[:bias.numel()] = bias
    return wp, bp


def _fused_outputs():
    """A/B switch: LFD_OUT_FUSED=0 runs the glue around the output convs as PyTorch ops (~25 launches per level)"""
    return os.environ.get('LFD_OUT_FUSED', '1') == '1'


def _out_segs(o):
    segs, r0 = [], 0
    for kind, conv in o.convs:
        segs.append(dict(kind=kind, conv=conv, channels=conv.out_channels, row0=r0,
                         scale=o.scale._scale.detach() if (kind == 'reg' and o.scale is not None) else None))
        r0 += conv.out_channels
    return segs


def _outputs_forward_fused(outs, acts, num_levels):
    """outputs_forward with one launch per level behind the padded conv (ops.head_out_split) writing straight into the
    level-concatenated tensors"""
    sizes = [None] * num_levels
    for o in outs:
        sizes[o.level] = tuple(acts[o.src].shape[1:3])
    starts, p = [], 0
    for h, w_ in sizes:
        starts.append(p)
        p += h * w_
    x0 = acts[outs[0].src]
    n = x0.size(0)
    width = {}
    for o in outs:
        for kind, conv in o.convs:
            width[kind] = conv.out_channels
    full = {k: torch.empty((n, p, c), dtype=torch.float32, device=x0.device) for k, c in width.items()}
    cache, saved = {}, []
    for o in outs:
        x = acts[o.src]
        c = x.size(3)
        key = tuple(id(cv) for _, cv in o.convs)
        if key not in cache:
            wp, bp = _out_weight(o)
            cache[key] = (wp, bp, ops.pack_conv_weight_train(wp))
        wp, bp, wpk = cache[key]
        y = ops.conv2d_nhwc(x, wpk, bp, c, wp.size(0), 1, 1, False)
        segs = _out_segs(o)
        ops.head_out_split(y, segs, [full[sg['kind']] for sg in segs], starts[o.level])
        saved.append((wp, y))
    return full['cls'], full['reg'], sizes, saved


def _outputs_backward_fused(outs, acts, saved, sizes, dcls, dreg, store, scale):
    grads = {}
    inv = 1.0 / scale
    starts, p = [], 0
    for h, w_ in sizes:
        starts.append(p)
        p += h * w_
    zeros = _Zeros(dcls.device)
    packs = _Packs()
    full = {'cls': dcls.contiguous(), 'reg': dreg.contiguous()}
    dws = {}
    for o, (wp, y) in zip(outs, saved):
        x = acts[o.src]
        c = x.size(3)
        segs = _out_segs(o)
        for sg in segs:
            sg['dbias'] = store.target(sg['conv'].bias)
            sg['dscale'] = store.target(o.scale._scale) if sg['scale'] is not None else None
        dy = ops.head_out_grad(y, segs, [full[sg['kind']] for sg in segs], starts[o.level], scale)
        key = tuple(id(cv) for _, cv in o.convs)
        if key not in dws:         # one padded weight-gradient buffer per set of (possibly shared) output convs
            dws[key] = (torch.zeros_like(wp), o)
        ops.conv_wgrad(x, dy, 1, 1, inv, out=dws[key][0], accumulate=True)
        grads[o.src] = ops.conv2d_nhwc(dy, packs(wp, True), zeros(c), wp.size(0), c, 1, 1, False, residual=grads.get(o.src))
    for dw, o in dws.values():
        r0 = 0
        for _, conv in o.convs:
            store.add(conv.weight, dw[r0:r0 + conv.out_channels])
            r0 += conv.out_channels
    return grads


def outputs_forward(outs, acts, num_levels):
    """-> (cls [N,P,C'], reg [N,P,4]) fp32 in the level-concatenated layout of LFD.forward (lfd.py:526-542), sizes per level,
    and what the backward needs."""
    if _fused_outputs():
        return _outputs_forward_fused(outs, acts, num_levels)
    cls_l, reg_l, sizes, saved = [None] * num_levels, [None] * num_levels, [None] * num_levels, []
    cache = {}
    for o in outs:
        x = acts[o.src]
        n, h, w_, c = x.shape
        key = tuple(id(cv) for _, cv in o.convs)           # shared heads: one concatenation + pack for all levels
        if key not in cache:
            wp, bp = _out_weight(o)
            cache[key] = (wp, bp, ops.pack_conv_weight_train(wp))
        wp, bp, wpk = cache[key]
        y = ops.conv2d_nhwc(x, wpk, bp, c, wp.size(0), 1, 1, False).view(n, h * w_, wp.size(0))
        r0 = 0
        raw = None
        for kind, conv in o.convs:
            t = y[..., r0:r0 + conv.out_channels].float()
            r0 += conv.out_channels
            if kind == 'cls':
                cls_l[o.level] = t
            else:
                raw = t
                reg_l[o.level] = t * o.scale._scale.detach() if o.scale is not None else t
        sizes[o.level] = (h, w_)
        saved.append((wp, raw))
    return torch.cat(cls_l, 1), torch.cat(reg_l, 1), sizes, saved


def outputs_backward(outs, acts, saved, sizes, dcls, dreg, store, scale=None):
    """-> {activation index: scaled fp16 gradient} for the tower outputs; parameter gradients go to `store`."""
    grads = {}
    scale = loss_scale() if scale is None else scale
    if saved and saved[0][1] is not None and saved[0][1].dtype == torch.float16:      # saved by _outputs_forward_fused
        return _outputs_backward_fused(outs, acts, saved, sizes, dcls, dreg, store, scale)
    inv = 1.0 / scale
    starts, p = [], 0
    for h, w_ in sizes:
        starts.append(p)
        p += h * w_
    zeros = _Zeros(dcls.device)
    packs = _Packs()
    for o, (wp, raw) in zip(outs, saved):
        x = acts[o.src]
        n, h, w_, c = x.shape
        lo, hi = starts[o.level], starts[o.level] + h * w_
        parts = []
        for kind, conv in o.convs:
            if kind == 'cls':
                d = dcls[:, lo:hi]
            else:
                d = dreg[:, lo:hi]
                if o.scale is not None:
                    store.add(o.scale._scale, (d * raw).sum())
                    d = d * o.scale._scale.detach()
            store.add(conv.bias, d.sum((0, 1)))
            parts.append(d)
        rows = wp.size(0)
        dy = torch.zeros((n, h, w_, rows), dtype=torch.float16, device=x.device)
        r0 = 0
        for d in parts:
            dy.view(n, h * w_, rows)[..., r0:r0 + d.size(2)] = (d * scale).half()
            r0 += d.size(2)
        dw = ops.conv_wgrad(x, dy, 1, 1, inv)
        r0 = 0
        for _, conv in o.convs:
            store.add(conv.weight, dw[r0:r0 + conv.out_channels])
            r0 += conv.out_channels
        grads[o.src] = ops.conv2d_nhwc(dy, packs(wp, True), zeros(c), rows, c, 1, 1, False, residual=grads.get(o.src))
    return grads


# ---------------------------------------------------------------------------------------------- whole-network schedule
# Round 4.  An iteration of the shipped configurations was ~500 launches of which ~360 ran for less than 12 us (small maps,
# one-wave finals): 2.3 ms of a 7.4 ms iteration as ONE dependent chain (tools/timing/train_trace.py).  What this schedule does
# about it, on ONE stream (side streams inside a HIP graph measured slower: tools/negative_results/train_streams.py.txt):
#   * a weight gradient is needed by nobody until the optimizer: every k_wgrad writes its per-workgroup partial sums into its
#     OWN persistent buffer and ONE launch sums all of them at the end (ops.WgradFinals) -- 49 final launches become one; a conv
#     shared by the pyramid levels is a chain of partial sets summed with one rounding;
#   * the shared head runs level-concatenated (below): 150 launches per iteration become ~70.


class _Sched(object):
    """persistent buffers and job tables of one training plan on one device"""

    def __init__(self, dev):
        self.dev = dev
        self.bufs = {}
        self.finals = ops.WgradFinals(dev)
        self.gather = ops.WgradFinals(dev)      # (its row-sum table with one source row = a batched copy)

    def buf(self, key, numel, dtype=torch.float32):
        """a buffer that keeps its address from iteration to iteration (partial sums of the weight gradients).  One buffer per
        (key, size): captured training graphs and the job tables of the batched finals hold these addresses, and a batch of
        another shape (the short last batch of an epoch goes through the eager train_step on the same model) must not free
        what a later replay writes into -- it gets its own buffers, both sets stay."""
        k = (key, int(numel), dtype)
        t = self.bufs.get(k)
        if t is None:
            t = torch.empty(numel, dtype=dtype, device=self.dev)
            self.bufs[k] = t
        return t


def _sched(plan_owner, dev):
    sc = plan_owner.__dict__.get('_lfd_train_sched')
    if sc is None or sc.dev != dev:
        sc = _Sched(dev)
        plan_owner.__dict__['_lfd_train_sched'] = sc
    return sc


def _out_packs(sc, outs):
    """{ids of a level's output convs: (padded weight [64, C, 1, 1], padded bias [64], forward pack, data-gradient pack)}: the
    rows of the convs are gathered into persistent zero-padded buffers by ONE launch (lfd_rows_sum_batched_f32 with one source
    row = a batched copy), then packed; shared heads: once for all levels"""
    cache, jobs = {}, []
    for o in outs:
        key = tuple(id(cv) for _, cv in o.convs)
        if key in cache:
            continue
        c = o.convs[0][1].in_channels
        rows = -(-sum(cv.out_channels for _, cv in o.convs) // 64) * 64
        wp = sc.bufs.get(('outw', key))
        if wp is None:
            wp = torch.zeros((rows, c, 1, 1), dtype=torch.float32, device=sc.dev)
            sc.bufs[('outw', key)] = wp
            sc.bufs[('outb', key)] = torch.zeros(rows, dtype=torch.float32, device=sc.dev)
        bp = sc.bufs[('outb', key)]
        r0 = 0
        for _, cv in o.convs:
            n_ = cv.out_channels
            jobs.append((cv.weight.detach(), wp[r0:r0 + n_]))
            jobs.append((cv.bias.detach(), bp[r0:r0 + n_]))
            r0 += n_
        cache[key] = [wp, bp]
    fin = sc.gather
    fin.reset()
    for src, dst in jobs:
        fin.add_rowsum(src, 1, src.numel(), src.numel(), dst, False)
    fin.launch()
    for v in cache.values():
        v.extend((ops.pack_conv_weight_train(v[0]), ops.pack_conv_weight_train(v[0], data_gradient=True)))
    return cache


def _out_pack(o, cache):
    return cache[tuple(id(cv) for _, cv in o.convs)]


# Level-concatenated head (round 4).  The shipped heads apply the SAME tower modules to every pyramid level (lfd_head.py:67-82,
# :164-185).  Per level that is 11 launches forward and 19 backward, 150 per iteration, most of them ~5 us kernels on the small
# levels.  With the levels of an image stored back to back -- [N, P, C], the point order LFD.forward returns anyway (lfd.py:526-542)
# -- a shared 1x1 conv, its weight gradient and its data gradient are ONE launch over all levels (a 1x1 conv does not care which
# pixel belongs to which level), GroupNorm takes its statistics per (image, level) segment (ops.gn_train_*_seg), and only the
# per-level modules stay per level: the neck units (own weights / BatchNorm; they write into / read from the concatenated tensor)
# and the glue of the output convs (per-level Scale).  Used when the head has this structure (every shipped configuration);
# CONCAT_HEAD = False, or a head whose levels do not share their towers, runs the same launches level by level.
CONCAT_HEAD = True


def _concat_layout(units, outs, nlev):
    """-> None, or the structure of a head whose towers and output convs are shared by all levels"""
    lv_units = [[ui for ui, u in enumerate(units) if u.level == l] for l in range(nlev)]
    k = len(lv_units[0]) if lv_units else 0
    if nlev < 2 or k < 2 or any(len(v) != k for v in lv_units):
        return None
    pos_of = [dict((units[ui].dst, p_) for p_, ui in enumerate(lv_units[l])) for l in range(nlev)]
    for p_ in range(k):
        u0 = units[lv_units[0][p_]]
        for l in range(nlev):
            u = units[lv_units[l][p_]]
            if u.res is not None or u.first or tuple(u.conv.kernel_size) != (1, 1) or tuple(u.conv.stride) != (1, 1) or not u.relu:
                return None
            if p_ == 0:      # the level's own neck unit on its backbone tap
                if not isinstance(u.norm, nn.BatchNorm2d) or u.src in pos_of[l] or u.conv.out_channels != u0.conv.out_channels:
                    return None
            else:
                if u.conv is not u0.conv or u.norm is not u0.norm or not isinstance(u.norm, nn.GroupNorm):
                    return None
                if pos_of[l].get(u.src) is None or pos_of[l].get(u.src) != pos_of[0].get(u0.src):
                    return None
    lv_outs = [[o for o in outs if o.level == l] for l in range(nlev)]
    m = len(lv_outs[0])
    if m < 1 or any(len(v) != m for v in lv_outs):
        return None
    for j in range(m):
        o0 = lv_outs[0][j]
        for l in range(nlev):
            o = lv_outs[l][j]
            if [id(cv) for _, cv in o.convs] != [id(cv) for _, cv in o0.convs]:
                return None
            if pos_of[l].get(o.src) is None or pos_of[l].get(o.src) != pos_of[0].get(o0.src):
                return None
    return dict(lv_units=lv_units, npos=k, src_pos=[None] + [pos_of[0][units[lv_units[0][p_]].src] for p_ in range(1, k)],
                out_src=[pos_of[0][lv_outs[0][j].src] for j in range(m)], lv_outs=lv_outs)


def _as_image(t):
    """[N, P, C] -> [1, H, W, C] view with N * P = H * W: how a 1x1 conv kernel sees the concatenated pixels (None: no usable W)"""
    total = t.size(0) * t.size(1)
    for w_ in (256, 128, 64, 32, 16):
        if total % w_ == 0:
            return t.view(1, total // w_, w_, t.size(2))
    return None


def _concat_forward(cl, units, outs, acts, tape, packs, opk, zeros, fused_stats, full, sizes, starts, n, dev, sc=None):
    """neck units per level into A[0]; shared tower units over all levels; output convs over all levels + per-level slices"""
    seg_hw = [h_ * w_ for h_, w_ in sizes]
    ptot = sum(seg_hw)
    A, Y = [None] * cl['npos'], [None] * cl['npos']
    c0 = units[cl['lv_units'][0][0]].conv.out_channels
    A[0] = torch.empty((n, ptot, c0), dtype=torch.float16, device=dev)
    if _as_image(A[0]) is None:
        return None
    # the neck units: independent of each other -- every conv leaves its statistics rows in a buffer of its own, then TWO launches
    # finish all levels (per-channel finals; apply passes into A[0]) instead of two per level (LFD_BN_LEVELS=0; the same values)
    pending = []
    batched = fused_stats and sc is not None and os.environ.get('LFD_BN_LEVELS', '1') == '1'
    for l in range(len(sizes)):
        ui = cl['lv_units'][l][0]
        u = units[ui]
        conv, norm = u.conv, u.norm
        xin = acts[u.src]
        if batched and all(id(norm) != id(units[q[1]].norm) for q in pending):      # (a norm shared by two levels: one after the other)
            rows = sc.buf(('bnrows', l), 512 * 2 * conv.out_channels)
            r = ops.conv2d_bn_partials(xin, packs(conv.weight), zeros(conv.out_channels), conv.in_channels, conv.out_channels, 1, 1, rows)
            if r is not None:
                pending.append((l, ui, (starts[l], r[0], rows, r[1], norm.eps, norm.momentum, norm.running_mean, norm.running_var,
                                        norm.weight.detach(), norm.bias.detach())))
                continue
        if fused_stats:
            y, stats = ops.conv2d_bn_stats(xin, packs(conv.weight), zeros(conv.out_channels), conv.in_channels, conv.out_channels, 1, 1,
                                           norm.eps, norm.momentum, norm.running_mean, norm.running_var)
        else:
            y = ops.conv2d_nhwc(xin, packs(conv.weight), zeros(conv.out_channels), conv.in_channels, conv.out_channels, 1, 1, False)
            stats = ops.bn_train_stats(y, norm.eps, norm.momentum, norm.running_mean, norm.running_var)
        ops.bn_train_apply_into(y, stats, norm.weight.detach(), norm.bias.detach(), True, A[0], starts[l])
        tape[ui] = (y, stats)
    if pending:
        sts = ops.bn_train_finish_into_levels([t[2] for t in pending], n, True, A[0])
        for (l, ui, lvl), st_ in zip(pending, sts):
            tape[ui] = (lvl[1], st_)
    for p_ in range(1, cl['npos']):
        u = units[cl['lv_units'][0][p_]]
        conv, norm = u.conv, u.norm
        xin = _as_image(A[cl['src_pos'][p_]])
        y = ops.conv2d_nhwc(xin, packs(conv.weight), zeros(conv.out_channels), conv.in_channels, conv.out_channels, 1, 1, False)
        y = y.view(n, ptot, conv.out_channels)
        stats, A[p_] = ops.gn_train_stats_apply_seg(y, seg_hw, norm.num_groups, norm.eps, norm.weight.detach(), norm.bias.detach(), True)
        Y[p_] = (y, stats)
    yo = []
    for j, sp in enumerate(cl['out_src']):
        o0 = cl['lv_outs'][0][j]
        wp, bp, wpk, _ = _out_pack(o0, opk)
        xin = _as_image(A[sp])
        y = ops.conv2d_nhwc(xin, wpk, bp, xin.size(3), wp.size(0), 1, 1, False).view(n, ptot, wp.size(0))
        lv = []
        for l in range(len(sizes)):
            segs = _out_segs(cl['lv_outs'][l][j])
            lv.append((seg_hw[l], starts[l], segs, [full[sg['kind']] for sg in segs]))
        if os.environ.get('LFD_HEAD_OUT_LEVELS', '1') == '1':
            ops.head_out_split_levels(y, lv)          # all levels (their own Scale each) in one launch
        else:
            for hw_, p0_, segs, outs_ in lv:
                ops.head_out_split_concat(y, hw_, segs, outs_, p0_)
        yo.append(y)
    return dict(A=A, Y=Y, yo=yo, seg_hw=seg_hw, ptot=ptot)


def _concat_backward(cl, cs, units, acts, tape, packs, opk, zeros, full, starts, store, wgrad, grads, scale, inv, n, dev):
    """the backward of _concat_forward; leaves the gradients of the backbone taps in `grads`"""
    A, Y, seg_hw, ptot = cs['A'], cs['Y'], cs['seg_hw'], cs['ptot']
    dA = [None] * cl['npos']
    nlev = len(seg_hw)
    for j, sp in enumerate(cl['out_src']):
        o0 = cl['lv_outs'][0][j]
        wp = _out_pack(o0, opk)[0]
        y = cs['yo'][j]
        dyo = torch.empty_like(y)
        lv = []
        for l in range(nlev):
            o = cl['lv_outs'][l][j]
            segs = _out_segs(o)
            for sg in segs:
                sg['dbias'] = store.target(sg['conv'].bias)
                sg['dscale'] = store.target(o.scale._scale) if sg['scale'] is not None else None
            lv.append((seg_hw[l], starts[l], segs, [full[sg['kind']] for sg in segs]))
        if os.environ.get('LFD_HEAD_OUT_LEVELS', '1') == '1':
            ops.head_out_grad_levels(y, lv, scale, dyo)        # all levels in one launch + one final
        else:
            for hw_, p0_, segs, grads_ in lv:
                ops.head_out_grad_concat(y, hw_, segs, grads_, p0_, scale, dyo)
        r0, targets = 0, []
        for _, cv in o0.convs:
            targets.append((store.target(cv.weight), r0, r0 + cv.out_channels))
            r0 += cv.out_channels
        xin, dy4 = _as_image(A[sp]), _as_image(dyo)
        wgrad(xin, dy4, 1, 1, targets)
        c = xin.size(3)
        res = _as_image(dA[sp]) if dA[sp] is not None else None
        dA[sp] = ops.conv2d_nhwc(dy4, _out_pack(o0, opk)[3], zeros(c), wp.size(0), c, 1, 1, False, residual=res).view(n, ptot, c)
    for p_ in range(cl['npos'] - 1, 0, -1):
        if dA[p_] is None:
            continue
        u = units[cl['lv_units'][0][p_]]
        conv, norm = u.conv, u.norm
        y, stats = Y[p_]
        dy = ops.gn_train_backward_seg(dA[p_], y, A[p_], seg_hw, norm.num_groups, stats, norm.weight.detach(), inv,
                                       store.target(norm.weight), store.target(norm.bias), True)
        dA[p_] = None
        sp = cl['src_pos'][p_]
        xin, dy4 = _as_image(A[sp]), _as_image(dy)
        wgrad(xin, dy4, 1, 1, [(store.target(conv.weight), 0, conv.out_channels)])
        cin = conv.in_channels
        res = _as_image(dA[sp]) if dA[sp] is not None else None
        dA[sp] = ops.conv2d_nhwc(dy4, packs(conv.weight, True), zeros(cin), conv.out_channels, cin, 1, 1, False,
                                 residual=res).view(n, ptot, cin)
    # the neck units' BatchNorm backward: the levels are independent of each other -- three launches for all of them
    # (LFD_BN_LEVELS=0: three per level; the same values)
    dys = None
    if os.environ.get('LFD_BN_LEVELS', '1') == '1':
        lv = []
        for l in range(nlev):
            u = units[cl['lv_units'][l][0]]
            y, stats = tape[cl['lv_units'][l][0]]
            lv.append((starts[l], y, stats, u.norm.weight.detach(), u.norm.bias.detach(), store.target(u.norm.weight),
                       store.target(u.norm.bias)))
        if len({t[5].data_ptr() for t in lv}) == len(lv):       # (levels sharing one norm module would race on its gradient)
            dys = ops.bn_train_backward_from_levels(dA[0], lv, inv, relu=True, accumulate=True)
    for l in range(nlev - 1, -1, -1):
        ui = cl['lv_units'][l][0]
        u = units[ui]
        conv, norm = u.conv, u.norm
        y, stats = tape[ui]
        dy = dys[l] if dys is not None else ops.bn_train_backward_from(
            dA[0], starts[l], y, stats, norm.weight.detach(), norm.bias.detach(), inv, store.target(norm.weight),
            store.target(norm.bias), relu=True, accumulate=True)
        xin = acts[u.src]
        wgrad(xin, dy, 1, 1, [(store.target(conv.weight), 0, conv.out_channels)])
        cin = conv.in_channels
        grads[u.src] = ops.conv2d_nhwc(dy, packs(conv.weight, True), zeros(cin), conv.out_channels, cin, 1, 1, False,
                                       residual=grads.get(u.src))


def _deferred_units(units, outs):
    """-> {index of a unit whose activation is never stored: index of its only consumer}.  A BatchNorm + ReLU unit without a
    residual whose output feeds exactly one 1x1 stride-1 conv unit (the first conv of each stem pair, lfd_resnet.py:376-413):
    the consumer normalises its operand inside the conv kernel (ops.conv1x1_of_bn_relu_bn_stats) and its weight gradient
    re-forms it (ops.conv1x1_wgrad_partials_of_bn_relu) -- one write and one read of the largest tensors of the iteration less.
    LFD_BN_APPLY_IN_CONV=0: every unit stores its activation."""
    if os.environ.get('LFD_BN_APPLY_IN_CONV', '1') != '1':
        return {}
    return _stem_pairs(units, outs)


def _stem_pairs(units, outs):
    """the structural part of _deferred_units: {producer unit index: index of its only consumer, a 1x1 stride-1 conv unit}"""
    out = {}
    for ui, vi in _single_consumers(units, outs).items():
        v = units[vi]
        if (isinstance(v.norm, nn.BatchNorm2d) and v.conv.kernel_size[0] == 1 and v.conv.stride[0] == 1
                and v.conv.in_channels == 64 and v.conv.out_channels in (64, 128)):     # (the kernel's two instances: ADVICE r4)
            out[ui] = vi
    return out


def _single_consumers(units, outs):
    """{index of a BatchNorm + ReLU unit without residual: index of the ONE unit that reads its activation} (backbone / neck
    units only; activations that are also pyramid taps, residual inputs or output-conv inputs do not qualify)"""
    uses = {}
    for vi, v in enumerate(units):
        uses.setdefault(v.src, []).append(vi)
        if v.res is not None:
            uses.setdefault(v.res, []).append(-1)
    for o in outs:
        uses.setdefault(o.src, []).append(-1)
    out = {}
    for ui, u in enumerate(units):
        if not (isinstance(u.norm, nn.BatchNorm2d) and u.relu and u.res is None and u.level is None):
            continue
        cons = uses.get(u.dst, [])
        if len(cons) != 1 or cons[0] < 0:
            continue
        v = units[cons[0]]
        if v.level is None and not v.first:
            out[ui] = cons[0]
    return out


def network_forward(model, plan, x):
    """-> (cls [N,P,C'], reg [N,P,4], sizes, saved): LFD.forward in train mode (lfd.py:511-542) over the unit schedule"""
    units, outs, nlev = plan
    dev = x.device
    sc = _sched(model, dev)
    acts, tape = {0: x}, [None] * len(units)
    zeros = _Zeros(dev)
    packs = _Packs(units, False)
    opk = _out_packs(sc, outs)
    hw = {0: (x.size(2), x.size(3))}
    for u in units:
        k, st_ = u.conv.kernel_size[0], u.conv.stride[0]
        h_, w_ = hw[u.src]
        hw[u.dst] = ((h_ + 2 * (k // 2) - k) // st_ + 1, (w_ + 2 * (k // 2) - k) // st_ + 1)
    sizes = [None] * nlev
    for o in outs:
        sizes[o.level] = hw[o.src]
    starts, p = [], 0
    for h_, w_ in sizes:
        starts.append(p)
        p += h_ * w_
    width = {}
    for o in outs:
        for kind, conv in o.convs:
            width[kind] = conv.out_channels
    full = {k: torch.empty((x.size(0), p, c), dtype=torch.float32, device=dev) for k, c in width.items()}
    fused_stats = os.environ.get('LFD_CONV_BN_STATS', '1') == '1'
    cl = None
    if CONCAT_HEAD:
        cl = model.__dict__.get('_lfd_concat_layout', 0)
        if cl == 0:
            cl = _concat_layout(units, outs, nlev)
            model.__dict__['_lfd_concat_layout'] = cl
    cs = None
    defer = _deferred_units(units, outs) if fused_stats else {}
    fed = {v: u for u, v in defer.items()}
    for ui, u in enumerate(units):
        if u.level is None:
            tape[ui] = _unit_forward(u, acts, packs, zeros, fused_stats, store=ui not in defer,
                                     producer=(units[fed[ui]], tape[fed[ui]]) if ui in fed else None)
    if cl is not None:
        cs = _concat_forward(cl, units, outs, acts, tape, packs, opk, zeros, fused_stats, full, sizes, starts, x.size(0), dev, sc)
    osaved = cs
    if cs is None:     # level by level: heads that do not share their towers, or no usable image view of N * P pixels
        for ui, u in enumerate(units):
            if u.level is not None:
                tape[ui] = _unit_forward(u, acts, packs, zeros, fused_stats)
        osaved = []
        for o in outs:
            xo = acts[o.src]
            c = xo.size(3)
            wp, bp, wpk, _ = _out_pack(o, opk)
            y = ops.conv2d_nhwc(xo, wpk, bp, c, wp.size(0), 1, 1, False)
            segs = _out_segs(o)
            ops.head_out_split(y, segs, [full[sg['kind']] for sg in segs], starts[o.level])
            osaved.append((wp, y))
    torch._foreach_add_([u.norm.num_batches_tracked for u in units if isinstance(u.norm, nn.BatchNorm2d)], 1)
    return full['cls'], full['reg'], sizes, ((acts, tape), osaved, opk)


def _unit_forward(u, acts, packs, zeros, fused_stats, store=True, producer=None):
    """conv -> norm (batch / group statistics) -> (+ residual) -> ReLU of one unit on the current stream; -> (y, stats).
    store=False: the activation is not written (_deferred_units: its consumer forms it from (y, stats)); producer = (unit,
    (y, stats)) of such an input."""
    conv, norm = u.conv, u.norm
    xin = acts[u.src]
    ks, st = conv.kernel_size[0], conv.stride[0]
    stats = None
    if producer is not None:
        pu, (py, pstats) = producer
        cout = conv.out_channels
        y, stats = ops.conv1x1_of_bn_relu_bn_stats(py, pstats, pu.norm.weight.detach(), pu.norm.bias.detach(), packs(conv.weight),
                                                   zeros(cout), cout, norm.eps, norm.momentum, norm.running_mean, norm.running_var)
    elif u.first and isinstance(norm, nn.BatchNorm2d) and fused_stats:
        y, stats = ops.stem_conv0_train_fwd_bn_stats(xin, conv.weight, norm.eps, norm.momentum, norm.running_mean,
                                                     norm.running_var)
    elif u.first:
        y = ops.stem_conv0_train_fwd(xin, conv.weight)
    elif isinstance(norm, nn.BatchNorm2d) and fused_stats:
        cout = conv.out_channels
        y, stats = ops.conv2d_bn_stats(xin, packs(conv.weight), zeros(cout), conv.in_channels, cout, ks, st, norm.eps,
                                       norm.momentum, norm.running_mean, norm.running_var)
    else:
        cout = conv.out_channels
        y = ops.conv2d_nhwc(xin, packs(conv.weight), zeros(cout), conv.in_channels, cout, ks, st, False)
    if isinstance(norm, nn.GroupNorm):
        stats, z = ops.gn_train_stats_apply(y, norm.num_groups, norm.eps, norm.weight.detach(), norm.bias.detach(), u.relu)
    else:
        if stats is None:
            stats = ops.bn_train_stats(y, norm.eps, norm.momentum, norm.running_mean, norm.running_var)
        z = ops.bn_train_apply(y, stats, norm.weight.detach(), norm.bias.detach(),
                               acts[u.res] if u.res is not None else None, u.relu) if store else None
    acts[u.dst] = z
    return y, stats


def network_backward(model, plan, saved, sizes, dcls, dreg, scale):
    """the backward of network_forward: parameter gradients accumulate into `.grad` (created zeroed when missing)."""
    units, outs, nlev = plan
    (acts, tape), osaved, opk = saved
    dev = dcls.device
    sc = _sched(model, dev)
    inv = 1.0 / scale
    store = _GradStore(in_place=True)
    fin = sc.finals
    fin.reset()
    zeros = _Zeros(dev)
    packs = _Packs(units, True)
    dcls, dreg = dcls.contiguous(), dreg.contiguous()
    full = {'cls': dcls, 'reg': dreg}
    starts, p = [], 0
    for h, w_ in sizes:
        starts.append(p)
        p += h * w_
    nj = [0]

    def wgrad(xin, dy, ks, st, targets, producer=None):
        """the partial sums of dW into this conv's own buffer; `targets` as ops.WgradFinals.add_wgrad.  producer = (unit, (y,
        stats)): xin is that unit's pre-normalisation output, its activation was never stored (_deferred_units)"""
        floats, nwg, nblk = ops.conv_wgrad_partial_floats(xin, dy, ks, st)
        part = sc.buf(('wg', nj[0]), floats)
        nj[0] += 1
        if producer is not None:
            pu, (_, pstats) = producer
            ops.conv1x1_wgrad_partials_of_bn_relu(xin, pstats, pu.norm.weight.detach(), pu.norm.bias.detach(), dy, part)
        else:
            ops.conv_wgrad_partials(xin, dy, ks, st, part)
        fin.add_wgrad(part, nwg, nblk, xin.size(3), dy.size(3), ks * ks, inv, targets)

    grads = {}
    # stem pairs (_stem_pairs, adjacent units only: nothing else may touch the training workspace in between): the 1x1 conv's
    # data gradient leaves the BatchNorm backward sums of the unit in front of it behind (LFD_BN_SUMS_IN_DGRAD=0: separate pass)
    fed = {}
    if os.environ.get('LFD_BN_SUMS_IN_DGRAD', '1') == '1' and os.environ.get('LFD_CONV_BN_STATS', '1') == '1':
        fed = {v: u_ for u_, v in _stem_pairs(units, outs).items()
               if v == u_ + 1 and units[u_].conv.out_channels == 64 and units[v].conv.out_channels == 64}
    sum_rows = {}
    concat = isinstance(osaved, dict)
    if concat:
        _concat_backward(model.__dict__['_lfd_concat_layout'], osaved, units, acts, tape, packs, opk, zeros, full, starts, store,
                         wgrad, grads, scale, inv, dcls.size(0), dev)
    # ---- output convs, level by level
    for o, (wp, y) in (() if concat else zip(outs, osaved)):
        xo = acts[o.src]
        c = xo.size(3)
        segs = _out_segs(o)
        for sg in segs:
            sg['dbias'] = store.target(sg['conv'].bias)
            sg['dscale'] = store.target(o.scale._scale) if sg['scale'] is not None else None
        dy = ops.head_out_grad(y, segs, [full[sg['kind']] for sg in segs], starts[o.level], scale)
        r0, targets = 0, []
        for _, cv in o.convs:
            targets.append((store.target(cv.weight), r0, r0 + cv.out_channels))
            r0 += cv.out_channels
        wgrad(xo, dy, 1, 1, targets)
        grads[o.src] = ops.conv2d_nhwc(dy, _out_pack(o, opk)[3], zeros(c), wp.size(0), c, 1, 1, False, residual=grads.get(o.src))
    # ---- conv / norm / ReLU units, last to first
    for ui in range(len(units) - 1, -1, -1):
        u = units[ui]
        if u.dst not in grads or (concat and u.level is not None):
            continue
        dz = grads.pop(u.dst)
        conv, norm = u.conv, u.norm
        y, stats = tape[ui]
        z = acts[u.dst] if u.relu else None
        g = None
        if (u.first and isinstance(norm, nn.BatchNorm2d) and u.relu and u.res is None and conv.out_channels == 64
                and os.environ.get('LFD_CONV0_BN_WGRAD', '1') == '1'):
            # the first unit has no data gradient: BatchNorm's sums, then the weight gradient straight from dz and y (no dy tensor)
            ops.stem_conv0_bn_bwd_wgrad(acts[u.src], dz, y, stats, norm.weight.detach(), norm.bias.detach(), inv,
                                        store.target(norm.weight), store.target(norm.bias), store.target(conv.weight),
                                        sum_rows=sum_rows.get(ui, 0))
            continue
        if isinstance(norm, nn.GroupNorm):
            dy = ops.gn_train_backward(dz, y, z, norm.num_groups, stats, norm.weight.detach(), inv, store.target(norm.weight),
                                       store.target(norm.bias), True)
        elif ui in sum_rows:
            dy = ops.bn_train_backward_rows(dz, y, stats, norm.weight.detach(), norm.bias.detach(), inv, store.target(norm.weight),
                                            store.target(norm.bias), sum_rows[ui])
        else:
            # without a residual input the ReLU mask is recomputed from y (one tensor less to read in both passes)
            dy, g = ops.bn_train_backward(dz, y, z if u.res is not None else None, stats, norm.weight.detach(), inv,
                                          store.target(norm.weight), store.target(norm.bias), want_g=u.res is not None,
                                          accumulate=True, relu=u.relu, beta=norm.bias.detach())
        if u.res is not None:
            grads[u.res] = g if u.res not in grads else grads[u.res] + g
        xin = acts[u.src]
        ks, st = conv.kernel_size[0], conv.stride[0]
        if u.first:
            ops.stem_conv0_wgrad(xin, dy, inv, out=store.target(conv.weight), accumulate=True)
            continue
        producer = None
        if xin is None:         # the input activation was never stored (_deferred_units): re-formed from its producer's y
            pi = [i for i, p_ in enumerate(units) if p_.dst == u.src][0]
            producer = (units[pi], tape[pi])
            xin = tape[pi][0]
        wgrad(xin, dy, ks, st, [(store.target(conv.weight), 0, conv.out_channels)], producer)
        cin = conv.in_channels
        if st == 2 and ks == 3 and cin == 64 and conv.out_channels == 64 and os.environ.get('LFD_DGRAD_S2', '1') == '1':
            # per output parity, 9 tap-products per 2 x 2 pixels instead of 36 and no zero-inserted tensor (csrc/dgrad_s2.hip)
            grads[u.src] = ops.conv3x3s2_dgrad(dy, packs(conv.weight, True), xin.size(1), xin.size(2), residual=grads.get(u.src))
        elif ui in fed and u.src not in grads:
            pu, (py, pstats) = units[fed[ui]], tape[fed[ui]]
            grads[u.src], sum_rows[fed[ui]] = ops.conv1x1_dgrad_bn_bwd_sums(dy, packs(conv.weight, True), zeros(cin), py, pstats,
                                                                            pu.norm.weight.detach(), pu.norm.bias.detach())
        else:
            if st == 2:
                dy = ops.zero_insert2(dy, xin.size(1), xin.size(2))
            grads[u.src] = ops.conv2d_nhwc(dy, packs(conv.weight, True), zeros(cin), conv.out_channels, cin, ks, 1, False,
                                           residual=grads.get(u.src))
    fin.launch()


class BackboneTrainFunction(torch.autograd.Function):
    """taps (NCHW fp32) = backbone(x) as one autograd node (used when neck / head are not covered by network_supported)."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        units, tap_ids = plan
        taps, saved = forward(units, tap_ids, x)
        ctx.plan, ctx.saved = plan, saved
        return tuple(t.permute(0, 3, 1, 2).float() for t in taps)

    @staticmethod
    def backward(ctx, *tap_grads):
        units, tap_ids = ctx.plan
        grads = {}
        scale = loss_scale()
        for t, g in zip(tap_ids, tap_grads):
            if g is not None:
                g16 = (g * scale).permute(0, 2, 3, 1).contiguous().half()
                grads[t] = g16 if t not in grads else grads[t] + g16
        backward(units, ctx.saved, grads, scale=scale, store=_GradStore(in_place=True))
        ctx.saved = None
        return (None, None) + (None,) * len(backbone_params(units))     # gradients were accumulated into .grad directly


class NetworkTrainFunction(torch.autograd.Function):
    """(cls [N,P,C'], reg [N,P,4]) = LFD.forward(x) in train mode as ONE autograd node: every conv / norm / ReLU of
    backbone, neck and head runs forward and backward on the hand-written kernels (network_forward / network_backward)."""

    @staticmethod
    def forward(ctx, model, plan, x, *params):
        cls, reg, sizes, saved = network_forward(model, plan, x)
        ctx.model, ctx.plan, ctx.saved, ctx.sizes = model, plan, saved, sizes
        ctx.shapes = (cls.shape, reg.shape)
        NetworkTrainFunction.last_sizes = sizes
        return cls, reg

    @staticmethod
    def backward(ctx, dcls, dreg):
        units, outs, _ = ctx.plan
        dev = ctx.saved[0][0][0].device
        if dcls is None:
            dcls = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=dev)
        if dreg is None:
            dreg = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=dev)
        network_backward(ctx.model, ctx.plan, ctx.saved, ctx.sizes, dcls, dreg, loss_scale())
        ctx.saved = None
        return (None, None, None) + (None,) * len(network_params(units, outs))   # accumulated into .grad directly


def backbone_train_forward(backbone, x):
    plan = backbone.__dict__.get('_lfd_train_plan')
    if plan is None:
        plan = build_units(backbone)
        backbone.__dict__['_lfd_train_plan'] = plan
    return BackboneTrainFunction.apply(plan, x, *backbone_params(plan[0]))


def network_train_forward(model, x):
    """-> (cls, reg, [(h, w)] per level)"""
    plan = model.__dict__.get('_lfd_train_plan')
    if plan is None:
        units, outs = build_network(model)
        plan = (units, outs, model._num_heads)
        model.__dict__['_lfd_train_plan'] = plan
    cls, reg = NetworkTrainFunction.apply(model, plan, x, *network_params(plan[0], plan[1]))
    return cls, reg, NetworkTrainFunction.last_sizes
