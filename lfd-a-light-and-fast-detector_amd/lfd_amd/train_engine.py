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
import torch
import torch.nn as nn

from . import ops

LOSS_SCALE = 1024.0   # power of two; activation gradients are stored as fp16 * LOSS_SCALE


class _Unit(object):
    __slots__ = ('conv', 'norm', 'relu', 'src', 'res', 'dst', 'first')


def supported(backbone):
    """The HIP training path covers BatchNorm2d + ReLU backbones with 32/64/128-channel convs, nothing frozen."""
    if backbone._norm_cfg is None or backbone._norm_cfg.get('type') != 'BatchNorm2d':
        return False
    if backbone._activation_cfg.get('type') != 'ReLU' or backbone._frozen_stages > 0 or backbone._norm_eval:
        return False
    if backbone._input_channels != 3:
        return False
    for m in backbone.modules():
        if isinstance(m, nn.Conv2d) and m.in_channels != 3:
            if m.in_channels not in (32, 64, 128) or m.out_channels not in (32, 64, 128):
                return False
        if isinstance(m, nn.BatchNorm2d) and (m.momentum is None or not m.affine or not m.track_running_stats):
            return False
    first = backbone._stem[0]
    return first.out_channels in (32, 64) and first.kernel_size == (3, 3) and first.stride == (2, 2)


def build_units(backbone):
    """-> (units, tap activation indices).  Activation 0 is the input image."""
    units, n_act = [], 1

    def add(conv, norm, relu, src, res=None):
        nonlocal n_act
        u = _Unit()
        u.conv, u.norm, u.relu, u.src, u.res, u.dst, u.first = conv, norm, relu, src, res, n_act, src == 0
        units.append(u)
        n_act += 1
        return u.dst

    cur = 0
    mods = list(backbone._stem)
    for i in range(0, len(mods), 3):          # (conv, norm, activation) triples
        cur = add(mods[i], mods[i + 1], True, cur)
    taps = {}
    want = [tuple(t) for t in backbone._out_indices]
    for si, nblk in enumerate(backbone._body_architecture):
        for bi in range(nblk):
            blk = getattr(backbone, 'stage%d' % si)[bi]
            ident = cur
            if blk._downsample is not None:
                ident = add(blk._downsample[0], blk._downsample[1], False, cur)
            a = cur
            for ci in range(1, blk.num_convs + 1):
                last = ci == blk.num_convs
                a = add(getattr(blk, '_conv%d' % ci), getattr(blk, '_norm%d' % ci), True, a, ident if last else None)
            cur = a
            if (si, bi) in want:
                taps[(si, bi)] = cur
    return units, [taps[t] for t in want]


def backbone_params(units):
    ps = []
    for u in units:
        ps += [u.conv.weight, u.norm.weight, u.norm.bias]
    return ps


def _dgrad_weight(w):
    """weights of the conv that maps dL/dy to dL/dx: swap the channel roles, flip the taps"""
    return ops.pack_conv_weight(w.detach().permute(1, 0, 2, 3).flip(2, 3))


def forward(units, tap_ids, x):
    """x: NCHW fp32 image batch (as LFD.forward receives it, lfd.py:511).  -> (tap tensors NHWC fp16, tape)."""
    acts = {0: x}
    tape = []
    dev = x.device
    zero_bias = {}
    for u in units:
        conv, norm = u.conv, u.norm
        xin = acts[u.src]
        ks, st = conv.kernel_size[0], conv.stride[0]
        if u.first:
            y = ops.stem_conv0_train_fwd(xin, conv.weight)
        else:
            cout = conv.out_channels
            if cout not in zero_bias:
                zero_bias[cout] = torch.zeros(cout, dtype=torch.float32, device=dev)
            y = ops.conv2d_nhwc(xin, ops.pack_conv_weight(conv.weight), zero_bias[cout], conv.in_channels, cout, ks, st,
                                False)
        stats = ops.bn_train_stats(y, norm.eps, norm.momentum, norm.running_mean, norm.running_var)
        z = ops.bn_train_apply(y, stats, norm.weight.detach(), norm.bias.detach(),
                               acts[u.res] if u.res is not None else None, u.relu)
        acts[u.dst] = z
        tape.append((y, stats))
    torch._foreach_add_([u.norm.num_batches_tracked for u in units], 1)
    return [acts[t] for t in tap_ids], (acts, tape)


def backward(units, tap_ids, saved, tap_grads, scale=LOSS_SCALE, trace=None):
    """tap_grads: dL/dtap, NHWC fp16 already multiplied by `scale` (None for unused taps).
    -> list of fp32 parameter gradients in backbone_params(units) order (None where nothing flowed).
    trace: optional list that receives the per-unit tensors (tests check every unit against PyTorch given the same
    inputs)."""
    acts, tape = saved
    inv = 1.0 / scale
    grads = {}
    for t, g in zip(tap_ids, tap_grads):
        if g is not None:
            grads[t] = g if t not in grads else grads[t] + g
    out = [None] * (3 * len(units))
    dev = acts[0].device
    zero_bias = {}
    for ui in range(len(units) - 1, -1, -1):
        u = units[ui]
        dz = grads.pop(u.dst, None)
        if dz is None:
            continue
        conv, norm = u.conv, u.norm
        y, stats = tape[ui]
        z = acts[u.dst] if u.relu else None
        dgamma = torch.empty_like(norm.weight, dtype=torch.float32)
        dbeta = torch.empty_like(norm.bias, dtype=torch.float32)
        dy, g = ops.bn_train_backward(dz, y, z, stats, norm.weight.detach(), inv, dgamma, dbeta, want_g=u.res is not None)
        if u.res is not None:
            grads[u.res] = g if u.res not in grads else grads[u.res] + g
        xin = acts[u.src]
        ks, st = conv.kernel_size[0], conv.stride[0]
        rec = dict(ui=ui, dz=dz, dy=dy, g=g, dx_prev=grads.get(u.src), dx=None) if trace is not None else None
        if u.first:
            dw = ops.stem_conv0_wgrad(xin, dy, inv)
        else:
            dw = ops.conv_wgrad(xin, dy, ks, st, inv)
            if st == 2:
                dy = ops.zero_insert2(dy, xin.size(1), xin.size(2))
            cin = conv.in_channels
            if cin not in zero_bias:
                zero_bias[cin] = torch.zeros(cin, dtype=torch.float32, device=dev)
            grads[u.src] = ops.conv2d_nhwc(dy, _dgrad_weight(conv.weight), zero_bias[cin], conv.out_channels, cin, ks, 1,
                                           False, residual=grads.get(u.src))
            if rec is not None:
                rec['dx'] = grads[u.src]
        if rec is not None:
            rec.update(dw=dw, dgamma=dgamma, dbeta=dbeta)
            trace.append(rec)
        out[3 * ui], out[3 * ui + 1], out[3 * ui + 2] = dw, dgamma, dbeta
    return out


class BackboneTrainFunction(torch.autograd.Function):
    """taps (NCHW fp32, what the torch neck consumes) = backbone(x); one autograd node for the whole backbone."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        units, tap_ids = plan
        taps, saved = forward(units, tap_ids, x)
        ctx.plan, ctx.saved = plan, saved
        return tuple(t.permute(0, 3, 1, 2).float() for t in taps)

    @staticmethod
    def backward(ctx, *tap_grads):
        units, tap_ids = ctx.plan
        gs = [None if g is None else (g * LOSS_SCALE).permute(0, 2, 3, 1).contiguous().half() for g in tap_grads]
        pg = backward(units, tap_ids, ctx.saved, gs)
        ctx.saved = None
        return (None, None) + tuple(pg)


def backbone_train_forward(backbone, x):
    plan = backbone.__dict__.get('_lfd_train_plan')
    if plan is None:
        plan = build_units(backbone)
        backbone.__dict__['_lfd_train_plan'] = plan
    return BackboneTrainFunction.apply(plan, x, *backbone_params(plan[0]))
