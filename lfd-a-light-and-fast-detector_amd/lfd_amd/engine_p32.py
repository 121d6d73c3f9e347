"""'fp32_storage' precision mode of the LFD eval forward -- a SHIPPED mode of the product, selected with
`LFD.precision = 'fp32_storage'` (default 'fp16' = engine.py).

BASELINE.json's north_star asks for cls / bbox tensors within 1e-3 of the reference's fp32 PyTorch path
(lfd/model/lfd.py:511-542).  The fp16 pipeline is at 1.3e-3 .. 2.3e-3 on sigma(cls) / sigma(reg): the rounding of fp16
weights and stored activations.  This mode keeps every inter-layer tensor fp32 NHWC in HBM and runs every
convolution as ONE launch of `lfd_p32_conv2d_nhwc_f32` (csrc/precise.hip): operands split exactly into fp16 hi + 2^-11 lo
parts inside the kernel, three MFMAs per k-step, fp32 accumulation, fp32 epilogue (bias + residual + Scale + ReLU);
GroupNorm + ReLU by `lfd_p32_groupnorm_relu_f32` (fp64 statistics).  No torch.nn.functional, no arithmetic on activations
outside liblfd_hip.so; the host side is what engine.py's is: BatchNorm fold (fp64 -> fp32, once per parameter version),
weight packing, buffer plumbing, optional HIP-graph capture.

Measured (tests/test_gpu_precise.py): raw logits within 1e-5 of the fp32 oracle at the BASELINE configs' own
shapes; cost: bench.py `precise` key.  Decode + NMS are the same fp32 kernels as in fp16 mode (ops.detect_batched).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, engine, ops
from ._lib import check, lib, ptr, stream_ptr

_UNION = engine._UNION
_LO = 2048.0


def pack_weight(w):
    """[cout, cin, k, k] fp32 (cin a multiple of 32) -> fp16 [slab][chunk][tap][kstep][hi | 2^11 lo][64 lanes][8]
    (csrc/precise.hip / include/lfd_hip.h lfd_p32_conv2d_nhwc_f32); rows zero-padded to a multiple of 32."""
    w = w.detach().float()
    cout, cin, ks, _ = w.shape
    assert cin % 32 == 0
    ns = -(-cout // 32)
    if ns * 32 != cout:
        w = torch.cat([w, w.new_zeros((ns * 32 - cout, cin, ks, ks))], 0)
    hi = w.half()
    lo = ((w - hi.float()) * _LO).half()

    def frag(t):   # -> [slab, chunk, tap, kk, khalf, co, j]
        t = t.reshape(ns, 32, cin // 32, 2, 2, 8, ks * ks)          # [slab, co, chunk, kk, khalf, j, tap]
        return t.permute(0, 2, 6, 3, 4, 1, 5)
    out = torch.stack([frag(hi), frag(lo)], 4)                       # [slab, chunk, tap, kk, hl, khalf, co, j]
    return out.reshape(ns, cin // 32, ks * ks, 2, 2, 64, 8).contiguous()


def _pad_bias(b):
    n = -(-b.numel() // 32) * 32
    out = b.new_zeros(n, dtype=torch.float32)
    out[:b.numel()] = b.float()
    return out


class _Op(object):
    __slots__ = ('kind', 'src', 'dst', 'res', 'cin', 'cout', 'ks', 'stride', 'relu', 'w', 'b', 'scale', 'patch', 'ref_w',
                 'gamma', 'beta', 'eps', 'groups', 'out', 'level', 'tail')


class PrecisePlan(object):
    """launch plan of the fp32-storage forward for one (module tree, parameter version)"""

    def __init__(self, model, device):
        self.device = device
        self.param_sig = engine._param_signature(model._backbone, model._neck, model._head)
        self._shape_cache = {}
        self.ops = []
        self.buf_channels = {}
        self.buf_scale = {}
        self._nbuf = 0
        with torch.no_grad():
            self._build(model)

    def _new_buf(self, channels, scale):
        b = self._nbuf
        self._nbuf += 1
        self.buf_channels[b], self.buf_scale[b] = channels, scale
        return b

    def _conv(self, src, w, b, ks, stride, relu, res=None, patch=False, scale=None, out=None, level=None, tail=None):
        """w: folded fp32 OIHW (channel-padded where engine.fold_conv_norm pads), b: fp32 bias;
        tail = (w2 [64,64,1,1], b2): a chained 1x1 + ReLU in the same launch (lfd_p32_conv2d_tail_nhwc_f32)"""
        dev = self.device
        o = _Op()
        o.tail = None if tail is None else (pack_weight(tail[0]).to(dev), _pad_bias(tail[1]).to(dev))
        o.kind, o.src, o.res, o.ks, o.stride, o.relu, o.patch, o.scale = 'conv', src, res, ks, stride, int(relu), patch, scale
        o.cout = w.shape[0]
        o.ref_w = w
        if patch:
            o.cin = 3
            w27 = w.permute(0, 2, 3, 1).reshape(w.shape[0], 27)
            w = torch.cat([w27, w27.new_zeros((w.shape[0], 5))], 1).reshape(w.shape[0], 32, 1, 1)
        else:
            o.cin = w.shape[1]
        o.w = pack_weight(w).to(dev)
        o.b = _pad_bias(b).to(dev)
        o.out, o.level = out, level
        if out is None:
            sc = (self.buf_scale[src] if src != 'input' else 1) * stride
            o.dst = self._new_buf(o.cout, sc)
        else:
            o.dst = None
        self.ops.append(o)
        return o.dst

    def _gn(self, buf, norm):
        o = _Op()
        o.kind, o.src = 'gn', buf
        o.gamma = norm.weight.detach().float().contiguous().to(self.device)
        o.beta = norm.bias.detach().float().contiguous().to(self.device)
        o.eps, o.groups = float(norm.eps), int(norm.num_groups)
        self.ops.append(o)

    def _build(self, model):
        bb, neck, head = model._backbone, model._neck, model._head
        if type(neck).__name__ != 'SimpleNeck' or type(head).__name__ != 'LFDHead':
            engine._unsupported("precision='fp32_storage' covers SimpleNeck + LFDHead (every shipped configuration)")
        if bb._input_channels != 3:
            engine._unsupported('input_channels != 3')
        if bb._norm_cfg is not None and bb._norm_cfg['type'] != 'BatchNorm2d':
            engine._unsupported('backbone norm must be BatchNorm2d (or None)')
        has_norm = bb._norm_cfg is not None
        step = 3 if has_norm else 2
        cur = 'input'
        spec = list(bb.stem_spec())
        folded = [engine.fold_conv_norm(bb._stem[i * step], bb._stem[i * step + 1] if has_norm else None) for i in range(len(spec))]
        import os
        fuse_tail = os.environ.get('LFD_P32_TAIL', '1') != '0'
        i = 0
        while i < len(spec):
            k, s, cin, cout = spec[i]
            w, b = folded[i]
            # a stem pair conv3x3 s2 -> conv1x1 (64 -> 64) runs as ONE launch: the fp32 intermediate stays in LDS
            tail = None
            if (fuse_tail and i + 1 < len(spec) and (k, s) == (3, 2) and w.shape[0] == 64 and spec[i + 1][0] == 1 and spec[i + 1][1] == 1
                    and tuple(folded[i + 1][0].shape[:2]) == (64, 64)):
                tail = folded[i + 1]
            if i == 0:
                if (k, s, cin) != (3, 2, 3):
                    engine._unsupported('first stem conv must be 3x3 stride 2 on 3 channels')
                cur = self._conv(cur, w, b, 3, 2, True, patch=True, tail=tail)
            else:
                cur = self._conv(cur, w, b, k, s, True, tail=tail)
            i += 2 if tail is not None else 1
        taps = [tuple(t) for t in bb._out_indices]
        self.taps = []
        for i, nblk in enumerate(bb._body_architecture):
            for j in range(nblk):
                blk = getattr(bb, 'stage%d' % i)[j]
                ident = cur
                if blk._downsample is not None:
                    dconv = blk._downsample[0]
                    w, b = engine.fold_conv_norm(dconv, blk._downsample[1] if len(blk._downsample) > 1 else None)
                    ident = self._conv(cur, w, b, dconv.kernel_size[0], dconv.stride[0], False)
                y = cur
                for ci in range(1, blk.num_convs + 1):
                    conv = getattr(blk, '_conv%d' % ci)
                    w, b = engine.fold_conv_norm(conv, getattr(blk, '_norm%d' % ci, None))
                    last = ci == blk.num_convs
                    # lfd_resnet.py:140-152: out = relu(norm_n(conv_n(...)) + identity)
                    y = self._conv(y, w, b, conv.kernel_size[0], conv.stride[0], True, res=ident if last else None)
                cur = y
                if (i, j) in taps:
                    self.taps.append(cur)
        # ---- neck + head (simple_neck.py:67-74, lfd_head.py:164-185)
        ncfg = head._norm_cfg
        has_hn = ncfg is not None
        lstep = 3 if has_hn else 2
        nl = head._num_conv_layers
        ks_h = head._conv_kernel_size
        union = head._regression_loss_type in _UNION
        self.cls_channels = head.num_cls_channels
        self.num_levels = head._num_heads

        def tower(seq, t):
            for l in range(nl):
                conv, norm = seq[l * lstep], (seq[l * lstep + 1] if has_hn else None)
                if isinstance(norm, nn.GroupNorm):
                    bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels)
                    t = self._conv(t, conv.weight.detach().float(), bias.to(conv.weight.device), ks_h, 1, False)
                    self._gn(t, norm)
                else:       # BatchNorm2d (eval: folded) or no norm
                    w, b = engine._fold_conv_norm(conv, norm)
                    t = self._conv(t, w, b, ks_h, 1, True)
            return t

        for i, f in enumerate(self.taps):
            nseq = getattr(neck, 'neck%d' % i)
            w, b = engine.fold_conv_norm(nseq[0], nseq[1] if neck._norm_cfg is not None else None)
            t = self._conv(f, w, b, 1, 1, True)
            cls_path = getattr(head, 'head%d_classification_path' % i)
            reg_path = getattr(head, 'head%d_regression_path' % i)
            scale = head._scales[i]._scale.detach().float().reshape(1).to(self.device) if union else None
            if head._merge_path_flag:
                tt = tower(getattr(head, 'head%d_merge_path' % i), t)
                cconv, rconv, tc, tr = cls_path[0], reg_path[0], tt, tt
            else:
                cconv, rconv = cls_path[nl * lstep], reg_path[nl * lstep]
                tc, tr = tower(cls_path, t), tower(reg_path, t)
            kf = cconv.kernel_size[0]
            self._conv(tc, cconv.weight.detach().float(), cconv.bias.detach().float(), kf, 1, False, out='cls', level=i)
            self._conv(tr, rconv.weight.detach().float(), rconv.bias.detach().float(), kf, 1, False, scale=scale, out='reg',
                       level=i)

    def state_for(self, n, h, w, slot=0):
        key = (n, h, w, slot)
        st = self._shape_cache.get(key)
        if st is None:
            st = _State(self, n, h, w)
            self._shape_cache[key] = st
        return st

    def run(self, x, fmt, st):
        l, sp = lib(), stream_ptr()
        for o in self.ops:
            if o.kind == 'gn':
                t = st.bufs[o.src]
                check(l.lfd_p32_groupnorm_relu_f32(ptr(t), st.n, t.shape[1] * t.shape[2], t.shape[3], o.groups, ptr(o.gamma),
                                                   ptr(o.beta), o.eps, 1, ptr(st.gn_ws), st.gn_ws.numel(), sp),
                      'lfd_p32_groupnorm_relu_f32')
                continue
            if o.patch:
                src, hh, ww, infmt = x, st.h, st.w, fmt
            else:
                src = st.bufs[o.src]
                hh, ww, infmt = src.shape[1], src.shape[2], -1
            d = _lib.P32ConvDesc(st.n, hh, ww, o.cin, o.cout, o.ks, o.stride, o.relu, infmt, 0, 0)
            if o.out is None:
                dst = ptr(st.bufs[o.dst])
            else:
                t = st.cls if o.out == 'cls' else st.reg
                cw = t.shape[2]
                d.out_pixel_stride, d.out_image_stride = cw, st.P * cw
                dst = C.c_void_p(t.data_ptr() + st.p_off[o.level] * cw * 4)
            if o.tail is not None:
                check(l.lfd_p32_conv2d_tail_nhwc_f32(C.byref(d), ptr(src), dst, ptr(o.w), ptr(o.b), ptr(o.tail[0]), ptr(o.tail[1]), 1, sp),
                      'lfd_p32_conv2d_tail_nhwc_f32')
                continue
            check(l.lfd_p32_conv2d_nhwc_f32(C.byref(d), ptr(src), dst, ptr(o.w), ptr(o.b),
                                            ptr(st.bufs[o.res]) if o.res is not None else None,
                                            ptr(o.scale) if o.scale is not None else None, sp), 'lfd_p32_conv2d_nhwc_f32')


class _State(object):
    """fp32 NHWC activation buffers and the [N,P,C'] / [N,P,4] outputs for one input shape"""

    def __init__(self, plan, n, h, w):
        dev = plan.device
        self.n, self.h, self.w = n, h, w
        self.bufs, self.dims = {}, {}
        with torch.cuda.device(dev):
            for b, sc in plan.buf_scale.items():
                hh, ww, s = h, w, sc
                while s > 1:
                    hh, ww = (hh + 1) // 2, (ww + 1) // 2
                    s //= 2
                self.dims[b] = (hh, ww)
                self.bufs[b] = torch.empty((n, hh, ww, plan.buf_channels[b]), dtype=torch.float32, device=dev)
            self.sizes = [self.dims[t] for t in plan.taps]
            self.p_off, p = [], 0
            for hh, ww in self.sizes:
                self.p_off.append(p)
                p += hh * ww
            self.P = p
            self.cls = torch.empty((n, p, plan.cls_channels), dtype=torch.float32, device=dev)
            self.reg = torch.empty((n, p, 4), dtype=torch.float32, device=dev)
            self.gn_ws = torch.empty(max(int(lib().lfd_p32_groupnorm_workspace_bytes(n, 64)), 1), dtype=torch.uint8, device=dev)
        self.graph = {}


def planes_enabled():
    """host-side A/B switch (tests, tools): LFD_P32_PLANES=0 keeps the fp32-tensor plan of this file"""
    import os
    return os.environ.get('LFD_P32_PLANES', '1') != '0'


def get_plan(model, device):
    """The hi/lo-plane plan (engine_p2.PlanesPlan, csrc/planes.hip) when every layer has a plane kernel -- all named
    configurations -- else the fp32-tensor plan of this file; same interface, same tolerance."""
    cache = model.__dict__.setdefault('_lfd_p32_cache', {})
    key = (device.type, device.index, planes_enabled())
    plan = cache.get(key)
    sig = engine._param_signature(model._backbone, model._neck, model._head)
    if plan is None or plan.param_sig != sig:
        plan = None
        if planes_enabled():
            from . import engine_p2
            try:
                plan = engine_p2.PlanesPlan(model, device)
            except engine_p2.Unsupported:
                plan = None
        if plan is None:
            plan = PrecisePlan(model, device)
        cache[key] = plan
    return plan


def lfd_forward(model, x, use_graph=False, slot=0):
    """(cls [N,P,C'] fp32, reg [N,P,4] fp32, sizes): engine-owned buffers, like engine.lfd_forward"""
    _lib.require_cuda(x, "LFD.forward (precision='fp32_storage')")
    if not x.is_contiguous():
        x = x.contiguous()
    fmt, n, h, w = engine._input_format(x)
    plan = get_plan(model, x.device)
    st = plan.state_for(n, h, w, slot)
    with torch.cuda.device(x.device):
        if not use_graph:
            plan.run(x, fmt, st)
        else:
            key = (x.data_ptr(), fmt)
            ent = st.graph.get(key)
            if ent is None:
                if len(st.graph) >= 4:
                    st.graph.pop(next(iter(st.graph)))
                plan.run(x, fmt, st)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    plan.run(x, fmt, st)
                ent = (g, x)
                st.graph[key] = ent
            ent[0].replay()
    return st.cls, st.reg, st.sizes
