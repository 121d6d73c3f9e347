"""G1 parity mode: the LFD eval forward with fp32 INTER-LAYER STORAGE, every convolution still on the shipped MFMA
conv kernels (SURVEY 8d "Parity gates", G1: kernels with fp32 inter-layer storage vs the fp32 oracle, <= 1e-4).

Purpose: separate the two things the fp16 pipeline's 1-2e-3 deviation from the fp32 reference could be made of --
(a) fp16 rounding of weights / stored activations (a numerics decision, measured in DESIGN 4) and (b) mistakes in the
math (BN fold, fragment packing, tile addressing, padding, stride, residual wiring, GroupNorm, Scale, level concat).
With (a) removed the whole network must agree with `oracle.net_oracle.lfd_forward` to ~1e-5.

How: an fp32 operand is split exactly into fp16 parts, x = x_hi + 2^-11 x_lo' (x_hi = fp16(x), x_lo' = fp16(2^11 (x -
x_hi)); the 2^11 keeps the low part out of the fp16 subnormals), and likewise the folded weights.  Every conv becomes three
launches of `lfd_conv2d_nhwc_f16_acc32` (conv_impl.h k_conv<..., ACC32>: the product kernel's tiling, LDS-DMA, swizzles
and v_mfma_f32_32x32x16_f16 contraction, with the fp32 accumulators written out un-rounded):

    conv(x, w) + b  =  acc32(x_hi, w_hi, b) + 2^-11 [ acc32(x_lo', w_hi) + acc32(x_hi, w_lo') ]     (+ O(2^-22))

Bias / residual add / ReLU / GroupNorm / Scale between the convs are fp32 tensor expressions on the device (debug-mode
glue: this path is an instrument, not the product -- engine.py never calls it; the fused stem and head kernels, which
round to fp16 INSIDE a launch, are covered in fp16 mode by the per-stage tests in tests/test_gpu_parity_fullsize.py).
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib, engine, ops
from ._lib import check, lib, ptr, stream_ptr

_LO = 2048.0    # 2^11
_UNION = ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss')


def _split(t):
    hi = t.half()
    lo = ((t - hi.float()) * _LO).half()
    return hi, lo


def _acc32(x16, w16_packed, bias, n, h, w, cin, cout, ks, stride):
    pad = ks // 2
    oh = (h + 2 * pad - ks) // stride + 1
    ow = (w + 2 * pad - ks) // stride + 1
    out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x16.device)
    d = _lib.ConvDesc(n, h, w, cin, cout, ks, stride, 0, 0, 0)
    check(lib().lfd_conv2d_nhwc_f16_acc32(C.byref(d), ptr(x16), ptr(out), ptr(w16_packed), ptr(bias),
                                          ptr(ops.zero_line(x16.device)), stream_ptr()), 'lfd_conv2d_nhwc_f16_acc32')
    return out


def conv_g1(x, w, b, stride):
    """x [N,H,W,Cin] fp32 NHWC (device), w [Cout,Cin,k,k] fp32 folded, b [Cout] fp32 -> conv + bias, fp32 NHWC."""
    n, h, wd, cin = x.shape
    cout, _, ks, _ = w.shape
    cin_p = max(32, -(-cin // 32) * 32) if cin not in (64, 128) else cin      # 3 -> 32
    cout_p = -(-cout // 32) * 32
    if cin == 128 and ks == 1 and cout_p == 32:
        cout_p = 64                                                             # instantiated: <128,1,1,NCT=2|4>
    if cin_p != cin:
        x = F.pad(x, (0, cin_p - cin))
        w = F.pad(w, (0, 0, 0, 0, 0, cin_p - cin))
    if cout_p != cout:
        w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_p - cout))
        b = F.pad(b, (0, cout_p - cout))
    x = x.contiguous()
    x_hi, x_lo = _split(x)
    w_hi, w_lo = _split(w.float())
    zero_b = torch.zeros_like(b)
    wp_hi, wp_lo = ops.pack_conv_weight(w_hi.float()), ops.pack_conv_weight(w_lo.float())
    b = b.float().contiguous()
    with torch.cuda.device(x.device):
        main = _acc32(x_hi, wp_hi, b, n, h, wd, cin_p, cout_p, ks, stride)
        c1 = _acc32(x_lo, wp_hi, zero_b, n, h, wd, cin_p, cout_p, ks, stride)
        c2 = _acc32(x_hi, wp_lo, zero_b, n, h, wd, cin_p, cout_p, ks, stride)
    out = main + (c1 + c2) * (1.0 / _LO)
    return out[..., :cout] if cout_p != cout else out


def _gn(x, norm):
    """GroupNorm on an NHWC fp32 tensor (statistics per image and group over H, W and the group's channels)."""
    y = F.group_norm(x.permute(0, 3, 1, 2), norm.num_groups, norm.weight, norm.bias, norm.eps)
    return y.permute(0, 2, 3, 1).contiguous()


@torch.no_grad()
def lfd_forward_g1(model, x):
    """x: NCHW fp32 on the device.  Returns (cls [N,P,C'] fp32, reg [N,P,4] fp32, sizes) like engine.lfd_forward."""
    _lib.require_cuda(x, 'lfd_forward_g1')
    bb, neck, head = model._backbone, model._neck, model._head
    has_norm = bb._norm_cfg is not None
    step = 3 if has_norm else 2
    y = x.float().permute(0, 2, 3, 1).contiguous()
    for i, (k, s, cin, cout) in enumerate(bb.stem_spec()):
        w, b = engine.fold_conv_norm(bb._stem[i * step], bb._stem[i * step + 1] if has_norm else None)
        y = conv_g1(y, w, b, s).relu_()
    feats = []
    taps = [tuple(t) for t in bb._out_indices]
    for i, nblk in enumerate(bb._body_architecture):
        for j in range(nblk):
            blk = getattr(bb, 'stage%d' % i)[j]
            ident = y
            if blk._downsample is not None:
                w, b = engine.fold_conv_norm(blk._downsample[0], blk._downsample[1] if len(blk._downsample) > 1 else None)
                ident = conv_g1(y, w, b, blk._downsample[0].stride[0])
            o = y
            for ci in range(1, blk.num_convs + 1):
                conv = getattr(blk, '_conv%d' % ci)
                w, b = engine.fold_conv_norm(conv, getattr(blk, '_norm%d' % ci, None))
                o = conv_g1(o, w, b, conv.stride[0])
                if ci < blk.num_convs:
                    o = o.relu_()
            y = (o + ident).relu_()
            if (i, j) in taps:
                feats.append(y)
    union = head._regression_loss_type in _UNION
    has_hn = head._norm_cfg is not None
    lstep = 3 if has_hn else 2
    cls_l, reg_l, sizes = [], [], []

    def tower(seq, t, nlayers):
        for l in range(nlayers):
            conv = seq[l * lstep]
            bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=t.device)
            t = conv_g1(t, conv.weight.detach().float(), bias, 1)
            if has_hn:
                nm = seq[l * lstep + 1]
                if isinstance(nm, torch.nn.GroupNorm):
                    t = _gn(t, nm)
                else:
                    t = F.batch_norm(t.permute(0, 3, 1, 2), nm.running_mean, nm.running_var, nm.weight, nm.bias, False, 0.0,
                                     nm.eps).permute(0, 2, 3, 1).contiguous()
            t = t.relu_()
        return t

    nl = head._num_conv_layers
    for i, f in enumerate(feats):
        nseq = getattr(neck, 'neck%d' % i)
        w, b = engine.fold_conv_norm(nseq[0], nseq[1] if neck._norm_cfg is not None else None)
        t = conv_g1(f, w, b, 1).relu_()
        cls_path = getattr(head, 'head%d_classification_path' % i)
        reg_path = getattr(head, 'head%d_regression_path' % i)
        if head._merge_path_flag:
            tt = tower(getattr(head, 'head%d_merge_path' % i), t, nl)
            cconv, rconv = cls_path[0], reg_path[0]
            c = conv_g1(tt, cconv.weight.detach().float(), cconv.bias.detach().float(), 1)
            r = conv_g1(tt, rconv.weight.detach().float(), rconv.bias.detach().float(), 1)
        else:
            cconv, rconv = cls_path[nl * lstep], reg_path[nl * lstep]
            c = conv_g1(tower(cls_path, t, nl), cconv.weight.detach().float(), cconv.bias.detach().float(), 1)
            r = conv_g1(tower(reg_path, t, nl), rconv.weight.detach().float(), rconv.bias.detach().float(), 1)
        if union:
            r = r * head._scales[i]._scale.detach().float()
        sizes.append((c.shape[1], c.shape[2]))
        cls_l.append(c.reshape(c.shape[0], -1, c.shape[3]))
        reg_l.append(r.reshape(r.shape[0], -1, 4))
    return torch.cat(cls_l, 1), torch.cat(reg_l, 1), sizes
