"""The 'fp32_storage' precision mode on hi/lo planes (csrc/planes.hip, csrc/planes_impl.h) -- the fast form of the mode inside
north_star's tolerance (lfd/model/lfd.py:511-542 to <= 1e-4 on the raw logits; tests/test_gpu_precise.py).

Every inter-layer tensor is a pair of NHWC fp16 planes, hi = fp16(x) and lo = fp16(2^11 (x - hi)) -- the bytes of the fp32
tensor csrc/precise.hip stores, in the form the matrix cores consume -- and the launches are the fused structures of the fp16
engine rebuilt on three MFMAs per k-step:

  stem        lfd_pl_stem_pair (frame -> conv3x3 s2 -> conv1x1) ; lfd_pl_conv2d 3x3 s2 + chained 1x1     lfd_resnet.py:376-413
  stage entry lfd_pl_conv2d 3x3 s2 with the 1x1 s2 identity branch as second output                       lfd_resnet.py:458-468
  block convs lfd_pl_conv2d 3x3 s1 (+ residual + ReLU)                                                    lfd_resnet.py:96-154
  head        neck 1x1 chained into the first tower 1x1 (GroupNorm sums from the epilogue); second tower 1x1 normalises +
              ReLUs the tile it fetched (+ its own sums); cls + reg 1x1 as ONE conv, again on the normalised tile, writing fp32
              [N,P,C'] / [N,P,4] (+ Scale): three launches per level, every tower tensor written once and read once
                                                                                        simple_neck.py:67-74, lfd_head.py:164-185

`PlanesPlan` has the interface of engine_p32.PrecisePlan (state_for / run); engine_p32.get_plan builds it first and falls
back to the fp32-tensor plan when a layer has no plane kernel (`Unsupported`).  The host side only folds BatchNorm (fp64),
splits and packs weights once per parameter version, and walks the launch list.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib, engine, ops
from ._lib import check, lib, ptr, stream_ptr

_UNION = engine._UNION
_LO = 2048.0


class Unsupported(Exception):
    pass


def _split(w):
    w = w.detach().float()
    hi = w.half()
    lo = ((w - hi.float()) * _LO).half()
    return hi, lo


def pack_planes_weight(w):
    """[cout, cin, k, k] fp32 -> fp16 [2 = hi | 2^11 lo][ceil(cout/32)][k*k*cin/16][64][8] (ops.pack_conv_weight order per
    plane, include/lfd_hip.h lfd_pl_conv2d); rows zero-padded to a multiple of 32."""
    w = w.detach().float()
    cout = w.shape[0]
    ns = -(-cout // 32)
    if ns * 32 != cout:
        w = torch.cat([w, w.new_zeros((ns * 32 - cout,) + tuple(w.shape[1:]))], 0)
    hi, lo = _split(w)
    return torch.stack([ops.pack_conv_weight(hi), ops.pack_conv_weight(lo)], 0).contiguous()


def pack_planes_stem_weight(w):
    """[C, 3, 3, 3] fp32 -> [2][C/32][2][64][8] (engine.pack_stem_weight order per plane)"""
    hi, lo = _split(w)
    return torch.stack([engine.pack_stem_weight(hi), engine.pack_stem_weight(lo)], 0).contiguous()


def pack_planes_stem2x_weight(w, b):
    """conv0 [64, 3, 3, 3] fp32 + bias [64] -> [2][2 slabs][2 k-steps][64 lanes][8] in the k-slot order of the fused stem's gather
    (csrc/planes_impl.h produce(): per frame row the aligned dwords [pixel left of the patch, e0..e8], e = 3 dx + c):
    step 0 {half 0: row 0 (junk, e0..e6), half 1: row 2 (junk, e0..e6)}, step 1 {half 0: row 1 (junk, e0..e6),
    half 1: (row 0 e7 e8, row 1 e7 e8, row 2 e7 e8, ONE, pad)}; junk / pad slots carry zero weights, the slot whose
    activation is the constant one carries the bias (split into hi / lo like a weight)."""
    if tuple(w.shape) != (64, 3, 3, 3) or b.numel() != 64:
        raise Unsupported('fused stem: conv0 must be [64, 3, 3, 3]')
    c = 64
    wr = w.detach().float().permute(0, 2, 3, 1).reshape(c // 32, 32, 3, 9)          # [slab][co][row][e = 3 dx + c]
    out = torch.zeros(c // 32, 2, 2, 32, 8, dtype=torch.float32, device=w.device)   # [slab][step][half][co][j]
    out[:, 0, 0, :, 1:] = wr[:, :, 0, :7]
    out[:, 0, 1, :, 1:] = wr[:, :, 2, :7]
    out[:, 1, 0, :, 1:] = wr[:, :, 1, :7]
    for r in range(3):
        out[:, 1, 1, :, 2 * r:2 * r + 2] = wr[:, :, r, 7:9]
    out[:, 1, 1, :, 6] = b.detach().float().reshape(c // 32, 32).to(w.device)
    hi, lo = _split(out.reshape(c // 32, 2, 64, 8))
    return torch.stack([hi, lo], 0).contiguous()


def pack_planes_stem2x_tail_weight(w):
    """the fused stem's first 1x1 [64, 64, 1, 1] -> [2][2 slabs][4 k-steps][64 lanes][8] with K in the order the conv0
    accumulators hold the channels: k-step q = 2 s + u, lane half h', element j -> input channel
    32 s + 16 u + (4 h' + j if j < 4 else 8 + 4 h' + j - 4)  (the 32x32 accumulator layout {8 g + 4 h + e}, g = 2 u + j // 4)"""
    if tuple(w.shape) != (64, 64, 1, 1):
        raise Unsupported('fused stem: first 1x1 must be [64, 64, 1, 1]')
    w2 = w.detach().float().reshape(64, 64)
    idx = torch.empty(4, 2, 8, dtype=torch.long)
    for q in range(4):
        s_, u = q // 2, q % 2
        for hh in range(2):
            for j in range(8):
                idx[q, hh, j] = 32 * s_ + 16 * u + (4 * hh + j if j < 4 else 8 + 4 * hh + j - 4)
    g = w2[:, idx.reshape(-1).to(w2.device)].reshape(2, 32, 4, 2, 8)          # [slab][co][q][h'][j]
    out = g.permute(0, 2, 3, 1, 4).reshape(2, 4, 64, 8)                      # [slab][q][lane = 32 h' + co][j]
    hi, lo = _split(out)
    return torch.stack([hi, lo], 0).contiguous()


def _flat_head_enabled():
    """LFD_P2_FLATHEAD=0: the neck / head through the generic multi-level conv (lfd_pl_conv2d_levels) instead of the flat-tile
    kernels of round 6 (lfd_pl_head_levels) -- A/B timing, tests"""
    return os.environ.get('LFD_P2_FLATHEAD', '1') != '0'


def _stem2x_enabled():
    """LFD_P2_STEM2X=0: the 'faster' stem as two launches (lfd_pl_stem_pair + lfd_pl_conv2d with a chained 1x1) instead of
    lfd_pl_stem2x (A/B timing, tests); =2: lfd_pl_stem2x for every input format (tests)"""
    return os.environ.get('LFD_P2_STEM2X', '1') != '0'


def _pad_bias(b, mult=32):
    n = -(-b.numel() // mult) * mult
    out = b.new_zeros(n, dtype=torch.float32)
    out[:b.numel()] = b.float()
    return out


def to_planes(x):
    """fp32 NHWC tensor -> planes [2, N, H, W, C] fp16 (tests, tools)"""
    hi = x.half()
    return torch.stack([hi, ((x - hi.float()) * _LO).half()], 0).contiguous()


def from_planes(p):
    return p[0].float() + p[1].float() / _LO


# (cin, ks, stride, ceil(cout / 32)) -> options csrc/planes.hip has an instance for (lfd_pl_conv2d's dispatch)
_DISPATCH = {
    (64, 3, 1, 2): {'res'}, (64, 3, 2, 2): {'tail', 'ds'}, (64, 3, 2, 4): {'ds'}, (64, 1, 1, 4): {'tail', 'gn', 'out32'},
    (128, 3, 1, 4): {'res'}, (128, 3, 2, 4): {'ds'}, (128, 1, 1, 4): {'tail', 'gn', 'out32'}, (128, 1, 1, 1): {'out32'},
    (128, 1, 1, 2): {'out32'}, (32, 3, 2, 1): {'tail', 'ds'}, (32, 3, 2, 2): {'tail', 'ds'},
}


def _check_dispatch(cin, cout, ks, stride, tail, ds, res, gn, out32):
    opts = _DISPATCH.get((cin, ks, stride, -(-cout // 32)))
    used = {k for k, v in (('tail', tail), ('ds', ds), ('res', res), ('gn', gn), ('out32', out32)) if v}
    if opts is None or not (used - {'gn'} <= opts and (not gn or 'gn' in opts)) or len(used - {'gn'}) > 1:
        raise Unsupported('no plane kernel for conv %d -> %d k%d s%d %s' % (cin, cout, ks, stride, sorted(used)))
    if gn and (res or ds or out32):
        raise Unsupported('GroupNorm sums ride on a plain or chained 1x1')


def _fork_enabled():
    import os
    return os.environ.get('LFD_P2_FORK', '0') != '0'


def _levels_enabled():
    """LFD_P2_LEVELS=0: one launch per pyramid level and head conv (A/B timing, tests) instead of one per conv over all levels"""
    return os.environ.get('LFD_P2_LEVELS', '1') != '0'


class _Op(object):
    __slots__ = ('kind', 'src', 'dst', 'res', 'ds_dst', 'cin', 'cout', 'ks', 'stride', 'relu', 'w', 'b', 'tail', 'ds',
                 'out_mode', 'gn', 'gnin', 'level', 'f_c0', 'f_c1', 'scale', 'w1', 'b1', 'w2', 'b2', 'channels')

    def __init__(self, kind):
        self.kind = kind
        for k in self.__slots__[1:]:
            setattr(self, k, None)


class PlanesPlan(object):
    """launch plan of the planes forward for one (module tree, parameter version)"""

    def __init__(self, model, device):
        self.device = device
        self.param_sig = engine._param_signature(model._backbone, model._neck, model._head)
        self._shape_cache = {}
        self.ops = []
        self.buf_channels = {}
        self.buf_scale = {}
        self._nbuf = 0
        self.num_gn = 0
        self.head_start = None      # index into self.ops of the first neck / head launch
        self.level_ops = []         # [(first, last + 1)] launch ranges of the pyramid levels' neck + head
        self.tap_ready = []         # index of the backbone launch that produces level i's input
        with torch.no_grad():
            self._build(model)

    # ------------------------------------------------------------------ plan construction
    def _new_buf(self, channels, scale):
        b = self._nbuf
        self._nbuf += 1
        self.buf_channels[b], self.buf_scale[b] = channels, scale
        return b

    def _conv(self, src, w, b, ks, stride, relu, res=None, tail=None, ds=None, gn=False, out32=None, gnin=None):
        """w: folded fp32 OIHW, b: fp32 bias.  tail = (w2 [c,c,1,1], b2, relu2); ds = (w [cout,cin,1,1], b): identity branch
        as second output; gn: GroupNorm sums of the stored values; out32 = (level, c0, c1, scale): fp32 cls / reg outputs;
        gnin = (sums index, GroupNorm module): `src` holds the pre-normalisation output of the conv with that sums index --
        GroupNorm + ReLU are applied to the landed tile inside this launch"""
        dev = self.device
        cout, cin = w.shape[0], w.shape[1]
        if cin not in (32, 64, 128) or (out32 is None and cout % 32):
            raise Unsupported('conv %d -> %d' % (cin, cout))
        _check_dispatch(cin, cout, ks, stride, tail is not None, ds is not None, res is not None, gn, out32 is not None)
        o = _Op('conv')
        if gnin is not None:
            norm = gnin[1]
            if not (isinstance(norm, nn.GroupNorm) and norm.num_channels == 128 and norm.num_groups == 16 and cin == 128 and ks == 1
                    and stride == 1 and tail is None and res is None and ds is None):
                raise Unsupported('head norm must be GroupNorm(16, 128) in front of a 1x1 conv')
            o.gnin = (gnin[0], norm.weight.detach().float().contiguous().to(dev), norm.bias.detach().float().contiguous().to(dev),
                      float(norm.eps))
        o.src, o.res, o.ks, o.stride, o.relu, o.cin, o.cout = src, res, ks, stride, int(relu), cin, cout
        o.w = pack_planes_weight(w).to(dev)
        o.b = _pad_bias(b, 128).to(dev)
        sc = self.buf_scale[src] * stride
        if tail is not None:
            if tuple(tail[0].shape) != (cout, cout, 1, 1):
                raise Unsupported('chained 1x1 must be square')
            o.tail = (pack_planes_weight(tail[0]).to(dev), _pad_bias(tail[1], 128).to(dev), int(tail[2]))
        if ds is not None:
            if tuple(ds[0].shape) != (cout, cin, 1, 1) or ks != 3 or stride != 2:
                raise Unsupported('identity branch must be the 1x1 stride-2 twin of a 3x3 stride-2 conv')
            o.ds = (pack_planes_weight(ds[0]).to(dev), _pad_bias(ds[1], 128).to(dev))
            o.ds_dst = self._new_buf(cout, sc)
        if out32 is not None:
            o.out_mode = 2
            o.level, o.f_c0, o.f_c1, o.scale = out32
        else:
            o.out_mode = 1 if gn else 0
            o.dst = self._new_buf(cout, sc)
            if gn:
                if cout != 128:
                    raise Unsupported('GroupNorm sums: 128 channels in groups of 8')
                o.gn = self.num_gn
                self.num_gn += 1
        self.ops.append(o)
        return o

    def _build(self, model):
        bb, neck, head = model._backbone, model._neck, model._head
        if type(neck).__name__ != 'SimpleNeck' or type(head).__name__ != 'LFDHead':
            raise Unsupported('SimpleNeck + LFDHead')
        if bb._input_channels != 3 or (bb._norm_cfg is not None and bb._norm_cfg['type'] != 'BatchNorm2d'):
            raise Unsupported('backbone')
        # the consumer kernels hard-wire ReLU (lfd_resnet.py / simple_neck.py / lfd_head.py build their activation from
        # activation_cfg): any other leaf module than conv / norm / ReLU / Scale falls back to the fp32-tensor plan's checks
        for part in (bb, neck, head):
            for mod in part.modules():
                if not any(True for _ in mod.children()) and type(mod).__name__ not in (
                        'Conv2d', 'BatchNorm2d', 'GroupNorm', 'ReLU', 'Scale', 'Identity', 'Sequential', 'ModuleList'):
                    raise Unsupported('module %s (activation must be ReLU)' % type(mod).__name__)
        has_norm = bb._norm_cfg is not None
        step = 3 if has_norm else 2
        spec = list(bb.stem_spec())
        folded = [engine.fold_conv_norm(bb._stem[i * step], bb._stem[i * step + 1] if has_norm else None) for i in range(len(spec))]
        # ---- stem: pairs conv3x3 s2 -> conv1x1 (lfd_resnet.py:356-413)
        if len(spec) not in (2, 4) or [(k, s) for k, s, _, _ in spec] != [(3, 2), (1, 1)] * (len(spec) // 2):
            raise Unsupported('stem mode')
        c = folded[0][0].shape[0]
        if c not in (32, 64) or tuple(folded[1][0].shape[:2]) != (c, c):
            raise Unsupported('stem channels')
        o = _Op('stem')
        o.channels = c
        o.w1, o.b1 = pack_planes_stem_weight(folded[0][0]).to(self.device), _pad_bias(folded[0][1]).to(self.device)
        o.w2, o.b2 = pack_planes_weight(folded[1][0]).to(self.device), _pad_bias(folded[1][1]).to(self.device)
        o.dst = self._new_buf(c, 2)
        self.ops.append(o)
        cur = o.dst
        self.stem2x = None
        if len(spec) == 4:
            if tuple(folded[3][0].shape[:2]) != (c, c):
                raise Unsupported('stem channels')
            pair2 = self._conv(cur, folded[2][0], folded[2][1], 3, 2, True, tail=(folded[3][0], folded[3][1], True))
            cur = pair2.dst
            if c == 64:
                # the whole stem as one launch (lfd_pl_stem2x) in place of ops[0:2]; the two-launch form stays for A/B
                f = _Op('stem2x')
                f.w1 = pack_planes_stem2x_weight(folded[0][0], folded[0][1]).to(self.device)
                f.w2, f.b2 = pack_planes_stem2x_tail_weight(folded[1][0]).to(self.device), o.b2
                f.dst = cur
                f.tail = pair2
                self.stem2x = f
        # ---- residual stages (lfd_resnet.py:96-154, :458-468, :488-501)
        taps = [tuple(t) for t in bb._out_indices]
        self.taps = []
        for i, nblk in enumerate(bb._body_architecture):
            for j in range(nblk):
                blk = getattr(bb, 'stage%d' % i)[j]
                ident, y = cur, cur
                for ci in range(1, blk.num_convs + 1):
                    conv = getattr(blk, '_conv%d' % ci)
                    w, b = engine.fold_conv_norm(conv, getattr(blk, '_norm%d' % ci, None))
                    ks, st = conv.kernel_size[0], conv.stride[0]
                    last = ci == blk.num_convs
                    ds = None
                    if ci == 1 and blk._downsample is not None:
                        dconv = blk._downsample[0]
                        if dconv.kernel_size[0] != 1 or dconv.stride[0] != 2 or (ks, st) != (3, 2) or last:
                            raise Unsupported('downsample branch')
                        ds = engine.fold_conv_norm(dconv, blk._downsample[1] if len(blk._downsample) > 1 else None)
                    elif ci > 1 and st != 1:
                        raise Unsupported('stride inside a block')
                    if ci == 1 and blk._downsample is None and st != 1:
                        raise Unsupported('strided block without a downsample branch')
                    op = self._conv(y, w, b, ks, st, True, res=ident if last else None, ds=ds)
                    y = op.dst
                    if ds is not None:
                        ident = op.ds_dst
                cur = y
                if (i, j) in taps:
                    self.taps.append(cur)
                    self.tap_ready.append(len(self.ops) - 1)
        # ---- neck + head (simple_neck.py:67-74, lfd_head.py:164-185)
        if head._norm_cfg is None or head._conv_kernel_size != 1:
            raise Unsupported('head towers must be 1x1 conv + GroupNorm')
        nl = head._num_conv_layers
        union = head._regression_loss_type in _UNION
        self.cls_channels = head.num_cls_channels
        self.num_levels = head._num_heads
        ccls = self.cls_channels

        def tower(seq, t, first_tail_of=None):
            """conv -> GroupNorm -> ReLU layers: every conv stores its pre-normalisation output + the GroupNorm sums, its
            consumer normalises the tile it has just fetched.  Returns (buffer, pending (sums index, norm)).  With
            `first_tail_of` = (neck w, b, src) the neck conv and the first tower conv are one launch."""
            pend = None
            for l in range(nl):
                conv, norm = seq[l * 3], seq[l * 3 + 1]
                if not isinstance(norm, nn.GroupNorm):
                    raise Unsupported('head norm must be GroupNorm')
                if not isinstance(seq[l * 3 + 2], nn.ReLU):
                    raise Unsupported('head activation must be ReLU')
                bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=conv.weight.device)
                w = conv.weight.detach().float()
                if l == 0 and first_tail_of is not None:
                    nw, nb, src = first_tail_of
                    op = self._conv(src, nw, nb, 1, 1, True, tail=(w, bias, False), gn=True)
                else:
                    op = self._conv(t, w, bias, 1, 1, False, gn=True, gnin=pend)
                pend = (op.gn, norm)
                t = op.dst
            return t, pend

        self.head_start = len(self.ops)
        for i, f in enumerate(self.taps):
            first_op = len(self.ops)
            nseq = getattr(neck, 'neck%d' % i)
            nw, nb = engine.fold_conv_norm(nseq[0], nseq[1] if neck._norm_cfg is not None else None)
            if nw.shape[0] != 128 or nw.shape[2] != 1:
                raise Unsupported('neck must be 1x1 -> 128')
            cls_path = getattr(head, 'head%d_classification_path' % i)
            reg_path = getattr(head, 'head%d_regression_path' % i)
            scale = head._scales[i]._scale.detach().float().reshape(1).to(self.device) if union else None
            if head._merge_path_flag:
                if nl < 1:
                    raise Unsupported('merge path without tower convs')
                tt, pend = tower(getattr(head, 'head%d_merge_path' % i), None, first_tail_of=(nw, nb, f))
                cconv, rconv = cls_path[0], reg_path[0]
                if cconv.kernel_size[0] != 1 or rconv.kernel_size[0] != 1 or ccls + 4 > 64:
                    raise Unsupported('output convs')
                w = torch.cat([cconv.weight.detach().float(), rconv.weight.detach().float()], 0)
                b = torch.cat([cconv.bias.detach().float(), rconv.bias.detach().float()], 0)
                self._conv(tt, w, b, 1, 1, False, out32=(i, ccls, 4, scale), gnin=pend)
            else:
                t = self._conv(f, nw, nb, 1, 1, True).dst
                cconv, rconv = cls_path[nl * 3], reg_path[nl * 3]
                if cconv.kernel_size[0] != 1 or rconv.kernel_size[0] != 1 or ccls > 64:
                    raise Unsupported('output convs')
                if nl < 1:
                    raise Unsupported('towers without convs')
                (tc, pc), (tr, pr) = tower(cls_path, t), tower(reg_path, t)
                self._conv(tc, cconv.weight.detach().float(), cconv.bias.detach().float(), 1, 1, False, out32=(i, ccls, 0, None), gnin=pc)
                self._conv(tr, rconv.weight.detach().float(), rconv.bias.detach().float(), 1, 1, False, out32=(i, 0, 4, scale), gnin=pr)
            self.level_ops.append((first_op, len(self.ops)))
        self.level_groups = self._group_levels()

    def _group_levels(self):
        """[[op index per level]]: the neck / head convs that run as ONE lfd_pl_conv2d_levels launch -- the k-th conv of every
        level's stack, split by launch signature (the neck's input channels differ between levels).  Position-major order keeps
        the producer of every GroupNorm sum in an earlier launch than its consumer."""
        lens = set(b - a for a, b in self.level_ops)
        if len(lens) != 1 or len(self.level_ops) > _lib.MAX_LEVELS:
            return None
        groups = []
        for k in range(lens.pop()):
            by_sig = {}
            for a, _ in self.level_ops:
                o = self.ops[a + k]
                if o.ks != 1 or o.stride != 1 or o.res is not None or o.ds is not None:
                    return None
                sig = (o.cin, o.cout, o.relu, o.tail is not None and o.tail[2], o.out_mode, o.gnin is not None and o.gnin[3],
                       o.f_c0, o.f_c1)
                by_sig.setdefault(sig, []).append(a + k)
            groups.extend(by_sig.values())
        return groups

    # ------------------------------------------------------------------ execution
    def state_for(self, n, h, w, slot=0):
        # (not evicted: the step graphs captured over a state replay raw addresses of its buffers -- model/lfd.py keeps them
        #  per frame buffer; a serving process sees a handful of input shapes)
        key = (n, h, w, slot)
        st = self._shape_cache.get(key)
        if st is None:
            st = _State(self, n, h, w)
            self._shape_cache[key] = st
        return st

    def run(self, x, fmt, st):
        """Launch order: backbone on the current stream; the neck + head of the FIRST pyramid level (the large map: a third
        of the head's work in three chip-filling launches) forks onto a side stream as soon as its input exists and runs
        beside the later backbone stages and the other levels' heads -- 30 short launches that each occupy a fraction of the
        CUs (small maps) -- and joins before the outputs are used.  Opt-in (LFD_P2_FORK=1): measured 1.905 vs 1.920 ms serial and 1.70 vs 1.675 ms
        with two batches in flight -- the fork of a captured graph starts late (DESIGN lesson 34) -- so one stream is the default."""
        if self.num_gn:
            st.gn_sums.zero_()
        fork = _fork_enabled() and len(self.level_ops) > 1 and self.tap_ready[0] + 1 < self.head_start
        if not fork:
            if self.level_groups is not None and _levels_enabled():
                self._launch(x, fmt, st, range(self.head_start))
                self._launch_levels(st)
            else:
                self._launch(x, fmt, st, range(len(self.ops)))
            return
        main = torch.cuda.current_stream()
        a, b = self.level_ops[0]
        self._launch(x, fmt, st, range(0, self.tap_ready[0] + 1))
        st.side.wait_stream(main)
        with torch.cuda.stream(st.side):
            self._launch(x, fmt, st, range(a, b))
        self._launch(x, fmt, st, list(range(self.tap_ready[0] + 1, self.head_start)) + list(range(b, len(self.ops))))
        main.wait_stream(st.side)

    def _desc(self, o, st):
        """(lfd_pl_conv_desc_t, src, dst, res, ds_dst, f_out0, f_out1) of conv op `o` on the buffers of `st`"""
        src = st.bufs[o.src]
        d = _lib.PlConvDesc()
        d.n, d.h, d.w, d.cin, d.cout, d.ks, d.stride, d.relu = st.n, src.shape[2], src.shape[3], o.cin, o.cout, o.ks, o.stride, o.relu
        d.out_mode = o.out_mode
        d.in_plane_halfs = src[0].numel()
        dst = res = dsd = None
        f0 = f1 = None
        if o.out_mode == 2:
            d.f_c0, d.f_c1 = o.f_c0, o.f_c1
            d.f_image_stride0, d.f_image_stride1 = st.P * o.f_c0, st.P * 4
            f0 = C.c_void_p(st.cls.data_ptr() + st.p_off[o.level] * o.f_c0 * 4) if o.f_c0 else None
            f1 = C.c_void_p(st.reg.data_ptr() + st.p_off[o.level] * 4 * 4) if o.f_c1 else None
        else:
            dst = st.bufs[o.dst]
            d.out_plane_halfs = dst[0].numel()
        if o.res is not None:
            res = st.bufs[o.res]
            d.res_plane_halfs = res[0].numel()
        if o.ds is not None:
            dsd = st.bufs[o.ds_dst]
            d.ds_plane_halfs = dsd[0].numel()
        if o.tail is not None:
            d.tail_cout, d.tail_relu = o.cout, o.tail[2]
        if o.gnin is not None:
            d.gn_in_eps = o.gnin[3]
        return d, src, dst, res, dsd, f0, f1

    def _flat_head_modes(self):
        """[mode per launch group] when every neck / head launch group is one of the forms lfd_pl_head_levels covers (round 6,
        csrc/planes_head.hip: flat pixel tiles, fp32 intermediates) or a plain neck conv (stays on lfd_pl_conv2d_levels: 'G') --
        neck + first tower conv (0), a tower conv on a GroupNorm input (1), the cls | reg output conv (2), a tower's first conv
        on the stored neck output (3: separate cls / reg towers, TT100K) -- else None."""
        if self.level_groups is None:
            return None
        modes = []
        for grp in self.level_groups:
            o = self.ops[grp[0]]
            plain = o.tail is None and o.gnin is None and o.res is None and o.ds is None
            if o.tail is not None and o.gn is not None and o.gnin is None and o.out_mode == 1 and o.cin in (64, 128) and o.cout == 128 and not o.tail[2]:
                modes.append(0)
            elif o.tail is None and o.gnin is not None and o.out_mode == 1 and o.cin == 128 and o.cout == 128:
                modes.append(1)
            elif o.tail is None and o.gnin is not None and o.out_mode == 2 and o.cin == 128 and o.f_c0 + o.f_c1 <= 64:
                modes.append(2)
            elif plain and o.gn is not None and o.out_mode == 1 and o.cin == 128 and o.cout == 128:
                modes.append(3)
            elif plain and o.gn is None and o.out_mode == 0:
                modes.append('G')
            else:
                return None
        return modes if any(m != 'G' for m in modes) else None

    def _flat_launches(self, modes):
        """[(mode, [op index])]: the launch groups in order, groups of the same flat mode merged into one launch where the level
        table has room and every producer of the later group's inputs sits in an EARLIER launch (the cls and reg towers of a
        separate-tower head: 4 levels x 2 towers = one launch per tower layer)"""
        launches, made_in = [], {}
        for grp, mode in zip(self.level_groups, modes):
            srcs = [self.ops[i].src for i in grp]
            target = None
            if mode in (1, 3):
                for li in range(len(launches) - 1, -1, -1):
                    m2, ops2 = launches[li]
                    if m2 == mode and len(ops2) + len(grp) <= _lib.MAX_LEVELS and all(made_in.get(b_, -1) < li for b_ in srcs):
                        target = li
                        break
                    if any(made_in.get(b_, -1) >= li for b_ in srcs):
                        break
            if target is None:
                launches.append((mode, list(grp)))
                target = len(launches) - 1
            else:
                launches[target][1].extend(grp)
            for i in grp:
                if self.ops[i].dst is not None:
                    made_in[self.ops[i].dst] = target
        return launches

    def _launch_levels_flat(self, st, modes):
        """the neck + head through lfd_pl_head_levels: the plane buffers of the tower convs hold fp32 [N, H, W, 128] (same bytes)"""
        l, sp = lib(), stream_ptr()
        zeros = ptr(ops.zero_line(self.device))
        if st.flat_calls is None:
            calls = []
            for mode, grp in self._flat_launches(modes):
                if mode == 'G':
                    arr = (_lib.PlLevel * len(grp))()
                    d0 = None
                    for j, i in enumerate(grp):
                        o = self.ops[i]
                        d, src, dst, _, _, _, _ = self._desc(o, st)
                        d0 = d0 or d
                        lv = arr[j]
                        lv.in_, lv.out, lv.w_packed, lv.bias = src.data_ptr(), dst.data_ptr(), o.w.data_ptr(), o.b.data_ptr()
                        lv.h, lv.w, lv.in_plane_halfs, lv.out_plane_halfs = src.shape[2], src.shape[3], d.in_plane_halfs, d.out_plane_halfs
                    calls.append(('G', d0, arr, len(grp)))
                    continue
                arr = (_lib.PlHeadLevel * len(grp))()
                d = _lib.PlHeadDesc()
                for j, i in enumerate(grp):
                    o = self.ops[i]
                    src = st.bufs[o.src]
                    lv, gi = arr[j], o.gnin
                    lv.in_, lv.w0, lv.b0 = src.data_ptr(), o.w.data_ptr(), o.b.data_ptr()
                    lv.pixels, lv.in_plane_halfs = src.shape[2] * src.shape[3], src[0].numel()
                    d.mode, d.n, d.cin, d.relu0 = mode, st.n, o.cin, o.relu
                    if mode == 0:
                        lv.w1, lv.b1 = o.tail[0].data_ptr(), o.tail[1].data_ptr()
                    if mode != 2:
                        lv.out, lv.gn_sums = st.bufs[o.dst].data_ptr(), st.gn_sums[o.gn].data_ptr()
                    if gi is not None:
                        lv.gn_in_sums, lv.gn_in_gamma, lv.gn_in_beta = st.gn_sums[gi[0]].data_ptr(), gi[1].data_ptr(), gi[2].data_ptr()
                        d.gn_in_eps = gi[3]
                    if mode == 2:
                        d.f_c0, d.f_c1 = o.f_c0, o.f_c1
                        d.f_image_stride0, d.f_image_stride1 = st.P * o.f_c0, st.P * 4
                        lv.f_out0 = (st.cls.data_ptr() + st.p_off[o.level] * o.f_c0 * 4) if o.f_c0 else None
                        lv.f_out1 = (st.reg.data_ptr() + st.p_off[o.level] * 4 * 4) if o.f_c1 else None
                        lv.scale1 = o.scale.data_ptr() if o.scale is not None else None
                calls.append((mode, d, arr, len(grp)))
            st.flat_calls = calls
        for mode, d, arr, n in st.flat_calls:
            if mode == 'G':
                check(l.lfd_pl_conv2d_levels(C.byref(d), arr, n, zeros, sp), 'lfd_pl_conv2d_levels')
            else:
                check(l.lfd_pl_head_levels(C.byref(d), arr, n, zeros, sp), 'lfd_pl_head_levels')

    def _launch_levels(self, st):
        """the neck + head: one launch per conv of the stack (and input width) over all pyramid levels"""
        modes = self._flat_head_modes() if _flat_head_enabled() else None
        if modes is not None:
            return self._launch_levels_flat(st, modes)
        l, sp = lib(), stream_ptr()
        zeros = ptr(ops.zero_line(self.device))
        if st.level_calls is None:
            calls = []
            for grp in self.level_groups:
                arr = (_lib.PlLevel * len(grp))()
                d0 = None
                for j, i in enumerate(grp):
                    o = self.ops[i]
                    d, src, dst, _, _, f0, f1 = self._desc(o, st)
                    d0 = d0 or d
                    lv, gi = arr[j], o.gnin
                    lv.in_, lv.out, lv.w_packed, lv.bias = src.data_ptr(), dst.data_ptr() if dst is not None else None, o.w.data_ptr(), o.b.data_ptr()
                    if o.tail is not None:
                        lv.tail_w_packed, lv.tail_bias = o.tail[0].data_ptr(), o.tail[1].data_ptr()
                    if o.gn is not None:
                        lv.gn_sums = st.gn_sums[o.gn].data_ptr()
                    if gi is not None:
                        lv.gn_in_sums, lv.gn_in_gamma, lv.gn_in_beta = st.gn_sums[gi[0]].data_ptr(), gi[1].data_ptr(), gi[2].data_ptr()
                    lv.f_out0, lv.f_out1 = f0, f1
                    lv.scale1 = o.scale.data_ptr() if o.scale is not None else None
                    lv.h, lv.w, lv.in_plane_halfs, lv.out_plane_halfs = src.shape[2], src.shape[3], d.in_plane_halfs, d.out_plane_halfs
                calls.append((d0, arr, len(grp)))
            st.level_calls = calls
        for d, arr, n in st.level_calls:
            check(l.lfd_pl_conv2d_levels(C.byref(d), arr, n, zeros, sp), 'lfd_pl_conv2d_levels')

    def _launch(self, x, fmt, st, indices):
        l, sp = lib(), stream_ptr()
        zeros = ptr(ops.zero_line(self.device))
        # the one-launch stem where its frame patches arrive by LDS-DMA (fp16 / uint8 NHWC, fp32 NCHW, 16-byte aligned rows: the row-stream
        # kernel); other layouts keep the two launches, whose loaders prefetch (lfd_pl_stem2x takes them too, by plain loads)
        fused = (self.stem2x is not None and _stem2x_enabled() and
                 (os.environ.get('LFD_P2_STEM2X') == '2' or
                  (x.data_ptr() % 16 == 0 and ((fmt == 1 and st.w % 8 == 0) or (fmt == 2 and st.w % 16 == 0) or (fmt == 0 and st.w % 4 == 0)))))
        for i in indices:
            o = self.ops[i]
            if fused and i < 2:
                if i == 0:
                    f, c2 = self.stem2x, self.stem2x.tail
                    dst = st.bufs[f.dst]
                    check(l.lfd_pl_stem2x(ptr(x), fmt, st.n, st.h, st.w, ptr(f.w1), ptr(f.w2), ptr(f.b2), ptr(c2.w), ptr(c2.b),
                                          ptr(c2.tail[0]), ptr(c2.tail[1]), ptr(dst), dst[0].numel(), zeros, sp), 'lfd_pl_stem2x')
                continue
            if o.kind == 'stem':
                dst = st.bufs[o.dst]
                check(l.lfd_pl_stem_pair(ptr(x), fmt, st.n, st.h, st.w, o.channels, ptr(o.w1), ptr(o.b1), ptr(o.w2), ptr(o.b2),
                                         ptr(dst), dst[0].numel(), sp), 'lfd_pl_stem_pair')
                continue
            d, src, dst, res, dsd, f0, f1 = self._desc(o, st)
            gi = o.gnin
            check(l.lfd_pl_conv2d(C.byref(d), ptr(src), ptr(dst), ptr(o.w), ptr(o.b), ptr(res),
                                  ptr(o.tail[0]) if o.tail else None, ptr(o.tail[1]) if o.tail else None,
                                  ptr(o.ds[0]) if o.ds else None, ptr(o.ds[1]) if o.ds else None, ptr(dsd),
                                  ptr(st.gn_sums[o.gn]) if o.gn is not None else None, f0, f1,
                                  ptr(o.scale) if o.scale is not None else None,
                                  ptr(st.gn_sums[gi[0]]) if gi else None, ptr(gi[1]) if gi else None, ptr(gi[2]) if gi else None,
                                  zeros, sp), 'lfd_pl_conv2d')


class _LazyBufs(dict):
    """plane buffer index -> tensor [2, N, H, W, C] fp16, allocated at its first use: a buffer no launch of the chosen launch
    list touches is never allocated -- the stem's pair-1 output (1.06 GB at 8 x 1080p) when the one-launch lfd_pl_stem2x runs
    (ADVICE r5).  First uses happen in the eager warm-up call in front of a graph capture (model/lfd.py), never inside one."""

    def __init__(self, st, plan):
        super().__init__()
        self._st, self._plan = st, plan

    def __missing__(self, b):
        hh, ww = self._st.dims[b]
        with torch.cuda.device(self._plan.device):
            t = torch.empty((2, self._st.n, hh, ww, self._plan.buf_channels[b]), dtype=torch.float16, device=self._plan.device)
        self[b] = t
        return t


class _State(object):
    """plane buffers [2, N, H, W, C] fp16, the fp32 [N,P,C'] / [N,P,4] outputs and the GroupNorm sums for one input shape"""

    def __init__(self, plan, n, h, w):
        dev = plan.device
        self.n, self.h, self.w = n, h, w
        self.dims = {}
        self.bufs = _LazyBufs(self, plan)
        with torch.cuda.device(dev):
            for b, sc in plan.buf_scale.items():
                hh, ww, s = h, w, sc
                while s > 1:
                    hh, ww = (hh + 1) // 2, (ww + 1) // 2
                    s //= 2
                self.dims[b] = (hh, ww)
            self.sizes = [self.dims[t] for t in plan.taps]
            self.p_off, p = [], 0
            for hh, ww in self.sizes:
                self.p_off.append(p)
                p += hh * ww
            self.P = p
            self.cls = torch.empty((n, p, plan.cls_channels), dtype=torch.float32, device=dev)
            self.reg = torch.empty((n, p, 4), dtype=torch.float32, device=dev)
            self.gn_sums = torch.zeros((max(plan.num_gn, 1), _lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device=dev)
            self.side = torch.cuda.Stream(device=dev)
        self.graph = {}
        self.level_calls = None
        self.flat_calls = None
