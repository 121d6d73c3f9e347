"""oracle/net_oracle.py -- TEST INFRASTRUCTURE.

Plain PyTorch fp32 CPU restatement (functional, driven by a reference-style state_dict)
of the floating-point part of the LFD hot path: backbone -> neck -> head -> [N,P,C] concat,
point grid, decode, result packing, target assignment and get_loss.  This is the "torch
fp32 reference of the same op" the HIP conv/loss kernels are compared with; the
integer/index-exact pieces (NMS) live in lfd_oracle.c.

Pinned against the REAL reference modules (imported in the build container) by
tests/golden/make_golden.py -> tests/golden/*.npz, checked in tests/test_oracle_golden.py.

`arch` is a plain dict with the reference constructors' kwargs:
  backbone: block_mode, stem_mode, stem_channels, body_architecture, body_channels,
            out_indices                                   (lfd_resnet.py:233-247)
  neck:     num_neck_channels                             (simple_neck.py:20-25)
  head:     num_classes, num_head_channels, num_conv_layers, conv_kernel_size, gn_groups,
            share_head_flag, merge_path_flag, classification_loss_type,
            regression_loss_type                          (lfd_head.py:32-45)
  lfd:      regression_ranges, point_strides (derived), distance_to_bbox_mode,
            range_assign_mode, gray_range_factors         (lfd.py:17-33)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle

BN_EPS = 1e-5   # nn.BatchNorm2d default, reference never overrides (lfd_resnet.py:10-18)
GN_EPS = 1e-5   # nn.GroupNorm default


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.0, BN_EPS)


def _conv(sd, prefix, x, stride=1, pad=0):
    return F.conv2d(x, sd[prefix + '.weight'], sd.get(prefix + '.bias'), stride=stride, padding=pad)


def strides_of(arch):
    """lfd_resnet.py:301-304."""
    stem_stride = 2 if arch['stem_mode'] == 'fast' else 4
    oi = sorted(arch['out_indices'])
    return [stem_stride * 2 ** (s + 1) for s, _ in oi]


def backbone_forward(sd, arch, x, pfx='_backbone.'):
    """LFDResNet.forward (lfd_resnet.py:488-501): stem (:354-439) then stages of blocks
    (:441-473); taps the (stage, block) outputs listed in out_indices (sorted :272)."""
    mode = arch['stem_mode']
    if mode == 'fast':        # conv3x3 s2, conv1x1   (:356-374)
        seq = [(0, 1, 2, 1), (3, 4, 1, 0)]
    elif mode == 'faster':    # conv3x3 s2, conv1x1, conv3x3 s2, conv1x1  (:376-413)
        seq = [(0, 1, 2, 1), (3, 4, 1, 0), (6, 7, 2, 1), (9, 10, 1, 0)]
    else:                     # 'fastest': conv3x3 s2 (C/2), conv3x3 s2  (:415-434)
        seq = [(0, 1, 2, 1), (3, 4, 2, 1)]
    for ci, ni, s, p in seq:
        x = F.relu(_bn(sd, f'{pfx}_stem.{ni}', _conv(sd, f'{pfx}_stem.{ci}', x, s, p)))
    outs = []
    oi = sorted(tuple(t) for t in arch['out_indices'])
    nstage = max(s for s, _ in oi) + 1
    bm = arch['block_mode']
    for i in range(nstage):
        for j in range(arch['body_architecture'][i]):
            b = f'{pfx}stage{i}.{j}.'
            stride = 2 if j == 0 else 1
            identity = x
            if bm == 'fast':      # FastBlock :21-93
                o = F.relu(_bn(sd, b + '_norm1', _conv(sd, b + '_conv1', x, stride, 1)))
                o = F.relu(_bn(sd, b + '_norm2', _conv(sd, b + '_conv2', o, 1, 0)))
                o = _bn(sd, b + '_norm3', _conv(sd, b + '_conv3', o, 1, 1))
            else:                 # FasterBlock :96-154 / FastestBlock :157-215
                o = F.relu(_bn(sd, b + '_norm1', _conv(sd, b + '_conv1', x, stride, 1)))
                o = _bn(sd, b + '_norm2', _conv(sd, b + '_conv2', o, 1, 1))
            if j == 0:            # downsample = conv1x1 s2 + norm  (:458-468)
                identity = _bn(sd, b + '_downsample.1', _conv(sd, b + '_downsample.0', x, 2, 0))
            x = F.relu(o + identity)
            if (i, j) in oi:
                outs.append(x)
    return outs


def neck_forward(sd, arch, feats, pfx='_neck.'):
    """SimpleNeck.forward (simple_neck.py:67-74): per level conv1x1 + BN + ReLU."""
    return [F.relu(_bn(sd, f'{pfx}neck{i}.1', _conv(sd, f'{pfx}neck{i}.0', f)))
            for i, f in enumerate(feats)]


def head_forward(sd, arch, feats, pfx='_head.'):
    """LFDHead.forward (lfd_head.py:164-185)."""
    G = arch['gn_groups']            # None: norm-free head (TrafficLight configs, lfd_head.py:100-117: conv(bias) + activation)
    k = arch.get('conv_kernel_size', 1)
    nl = arch.get('num_conv_layers', 2)
    ls = 3 if G else 2               # modules per tower layer: conv, [norm], activation

    def tower(name, x):
        for l in range(nl):
            x = _conv(sd, f'{name}.{ls * l}', x, 1, k // 2)
            if G:
                x = F.group_norm(x, G, sd[f'{name}.{ls * l + 1}.weight'], sd[f'{name}.{ls * l + 1}.bias'], GN_EPS)
            x = F.relu(x)
        return x

    cls_out, reg_out = [], []
    union = arch['regression_loss_type'] in ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss')
    for i, f in enumerate(feats):
        h = f'{pfx}head{i}_'
        if arch['merge_path_flag']:
            t = tower(h + 'merge_path', f)
            c = _conv(sd, h + 'classification_path.0', t)
            r = _conv(sd, h + 'regression_path.0', t)
        else:
            c = _conv(sd, h + f'classification_path.{ls * nl}', tower(h + 'classification_path', f))
            r = _conv(sd, h + f'regression_path.{ls * nl}', tower(h + 'regression_path', f))
        if union:
            r = r * sd[f'{pfx}_scales.{i}._scale']
        cls_out.append(c)
        reg_out.append(r)
    return cls_out, reg_out


def lfd_forward(sd, arch, x, return_intermediates=False):
    """LFD.forward (lfd.py:511-542): returns cls [N,P,C'], reg [N,P,4] and the per-level (h,w)."""
    feats = backbone_forward(sd, arch, x)
    necks = neck_forward(sd, arch, feats)
    cls_l, reg_l = head_forward(sd, arch, necks)
    sizes = [(c.shape[2], c.shape[3]) for c in cls_l]
    cls = torch.cat([c.permute(0, 2, 3, 1).reshape(c.shape[0], -1, c.shape[1]) for c in cls_l], 1)
    reg = torch.cat([r.permute(0, 2, 3, 1).reshape(r.shape[0], -1, 4) for r in reg_l], 1)
    if return_intermediates:
        return cls, reg, sizes, dict(backbone=feats, neck=necks, cls=cls_l, reg=reg_l)
    return cls, reg, sizes


# ---------------------------------------------------------------- post-processing
def point_coordinates(sizes, strides):
    """generate_point_coordinates (lfd.py:84-107): x=j*stride, y=i*stride, row-major, int64."""
    out = []
    for (h, w), s in zip(sizes, strides):
        ys, xs = np.meshgrid(np.arange(h, dtype=np.int64) * s, np.arange(w, dtype=np.int64) * s, indexing='ij')
        out.append(np.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    return out


def scores_from_logits(cls_logits, ce_loss):
    """lfd.py:450-454 / :584-588: softmax then drop bg for CrossEntropyLoss, else sigmoid."""
    t = torch.as_tensor(cls_logits, dtype=torch.float32)
    if ce_loss:
        return t.softmax(dim=1)[:, :-1].numpy()
    return t.sigmoid().numpy()


def decode_boxes(reg, sizes, strides, ranges, mode, loss_type, clamp_hw, resize_scale=1.0):
    """Decode in _get_results_for_single_image (lfd.py:468-499) / distance2bbox (:261-282).
    reg [P,4] fp32.  mode in {'sigmoid','exp'}, loss_type in {'union','independent'}."""
    reg = torch.as_tensor(reg, dtype=torch.float32)
    pts = torch.from_numpy(np.concatenate(point_coordinates(sizes, strides), 0))
    rmax = torch.cat([torch.full((h * w,), float(max(r)), dtype=torch.float32)
                      for (h, w), r in zip(sizes, ranges)])
    rhi = torch.cat([torch.full((h * w,), float(r[1]), dtype=torch.float32)
                     for (h, w), r in zip(sizes, ranges)])
    H, W = clamp_hw
    if loss_type == 'independent':
        d = reg * rhi[:, None]
    elif mode == 'exp':
        d = reg.float().exp()
    else:
        d = reg.sigmoid() * rmax[:, None]
    x1 = (pts[:, 0] - d[:, 0]).clamp(min=0, max=W)
    y1 = (pts[:, 1] - d[:, 1]).clamp(min=0, max=H)
    x2 = (pts[:, 0] + d[:, 2]).clamp(min=0, max=W)
    y2 = (pts[:, 1] + d[:, 3]).clamp(min=0, max=H)
    b = torch.stack([x1, y1, x2, y2], -1)
    if resize_scale != 1.0 or True:
        b = b / resize_scale          # lfd.py:499 (always divides, also by 1.0)
    return b.numpy()


def get_results_single(cls_logits, reg, sizes, strides, arch, score_thr, iou_thr,
                       class_agnostic, clamp_hw, resize_scale=1.0):
    """_get_results_for_single_image + packing (lfd.py:434-509, 418-431): returns
    dets[k,5] (x1,y1,x2,y2,score), labels[k], candidate ordinals[k], K."""
    ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
    lt = 'union' if arch['regression_loss_type'] in ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss') else 'independent'
    sc = scores_from_logits(cls_logits, ce)
    bx = decode_boxes(reg, sizes, strides, arch['regression_ranges'], arch['distance_to_bbox_mode'],
                      lt, clamp_hw, resize_scale)
    return c_oracle.multiclass_nms(bx, sc, score_thr, iou_thr, class_agnostic)


def pack_results(dets, labels):
    """lfd.py:421-429: [label, score, x1, y1, w=x2-x1+1, h=y2-y1+1]."""
    if dets.shape[0] == 0:
        return []
    d = dets.copy()
    d[:, 2] = d[:, 2] - d[:, 0] + 1
    d[:, 3] = d[:, 3] - d[:, 1] + 1
    return [[int(l), float(r[4]), float(r[0]), float(r[1]), float(r[2]), float(r[3])]
            for l, r in zip(labels, d)]


# ---------------------------------------------------------------- training targets / loss
def assign_targets_single(points, strides_pp, reg_ranges_pp, gray_ranges_pp, gt_bboxes, gt_labels,
                          num_classes, range_assign_mode='dist', loss_type='union'):
    """_generate_target_for_single_image (lfd.py:155-259), fp32, same op order.
    points [P,2] int64; strides_pp [P]; *_ranges_pp [P,2]; gt_bboxes [G,4] xywh f32; gt_labels [G] i64."""
    P = points.shape[0]
    G = gt_bboxes.shape[0]
    cls_t = torch.zeros((P, num_classes), dtype=torch.float32)
    reg_t = torch.zeros((P, 4), dtype=torch.float32)
    if G == 0:
        return cls_t, reg_t
    gb = gt_bboxes[None].expand(P, G, 4)
    gl = gt_labels[None].expand(P, G)
    rr = reg_ranges_pp[:, None, :].expand(P, G, 2)
    gr = gray_ranges_pp[:, None, :].expand(P, G, 2)
    px = points[:, 0][:, None].expand(P, G)
    py = points[:, 1][:, None].expand(P, G)
    cx = gb[..., 0] + gb[..., 2] / 2.
    cy = gb[..., 1] + gb[..., 3] / 2.
    st = strides_pp[:, None]
    xs = torch.abs(px - cx) / (st / 2.)
    xs = xs * (xs >= 1) + (xs < 1)
    xs = torch.sqrt(1. / xs)
    ys = torch.abs(py - cy) / (st / 2.)
    ys = ys * (ys >= 1) + (ys < 1)
    ys = torch.sqrt(1. / ys)
    score = xs * ys
    dx1 = px - gb[..., 0]
    dy1 = py - gb[..., 1]
    dx2 = (gb[..., 0] + gb[..., 2] - 1) - px
    dy2 = (gb[..., 1] + gb[..., 3] - 1) - py
    delta = torch.stack((dx1, dy1, dx2, dy2), dim=-1)
    if range_assign_mode == 'longer':
        measure = torch.max(gb[..., 2], gb[..., 3])
    elif range_assign_mode == 'shorter':
        measure = torch.min(gb[..., 2], gb[..., 3])
    elif range_assign_mode == 'sqrt':
        measure = torch.sqrt(gb[..., 2] * gb[..., 3])
    else:
        measure = delta.max(dim=-1)[0]
    if loss_type == 'independent':
        delta = delta / rr[..., 1, None]
    head_sel = (rr[..., 0] <= measure) & (measure <= rr[..., 1])
    hit = delta.min(dim=-1)[0] >= 0
    green = head_sel & hit
    g1 = (gr[..., 0] <= measure) & (measure < rr[..., 0])
    g2 = (rr[..., 1] < measure) & (measure <= gr[..., 1])
    gray = (g1 | g2) & hit
    sscore, sidx = score.sort(dim=1)
    rows = torch.arange(P)[:, None].expand(P, G)
    sl = gl[rows, sidx]
    sgreen = green[rows, sidx]
    sgray = gray[rows, sidx]
    i1, i2 = torch.where(sgreen)
    cls_t[i1, sl[i1, i2]] = sscore[i1, i2]
    i3, i4 = torch.where(sgray)
    cls_t[i3, sl[i3, i4]] = -1
    filt = sscore * (sgreen & ~sgray)
    _, sel = filt.max(dim=1)
    sdelta = delta[rows, sidx]
    reg_t = sdelta[torch.arange(P), sel]
    return cls_t, reg_t


def lfd_targets(arch, sizes, strides, gt_bboxes_list, gt_labels_list):
    """annotation_to_target (lfd.py:109-153)."""
    pts = point_coordinates(sizes, strides)
    points = torch.from_numpy(np.concatenate(pts, 0))
    gf = arch.get('gray_range_factors', (0.9, 1.1))
    gray = [(int(lo * min(gf)), int(up * max(gf))) for lo, up in arch['regression_ranges']]  # lfd.py:49-50
    st = torch.cat([torch.full((p.shape[0],), s, dtype=torch.int64) for p, s in zip(pts, strides)])
    rr = torch.cat([torch.tensor(r, dtype=torch.int64)[None].expand(p.shape[0], 2)
                    for p, r in zip(pts, arch['regression_ranges'])])
    gr = torch.cat([torch.tensor(r, dtype=torch.int64)[None].expand(p.shape[0], 2)
                    for p, r in zip(pts, gray)])
    lt = 'union' if arch['regression_loss_type'] in ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss') else 'independent'
    cl, rg = [], []
    for b, l in zip(gt_bboxes_list, gt_labels_list):
        c, r = assign_targets_single(points, st, rr, gr, torch.as_tensor(b, dtype=torch.float32),
                                     torch.as_tensor(l, dtype=torch.int64), arch['num_classes'],
                                     arch.get('range_assign_mode', 'dist'), lt)
        cl.append(c)
        rg.append(r)
    return torch.stack(cl), torch.stack(rg), points, rr


def union_box_loss(pred, target, kind, eps):
    """GIoU / DIoU / CIoU of aligned xyxy boxes [n,4] -> loss [n] (reference lfd/model/losses/iou_loss.py: giou_loss
    :127-169, diou_loss :172-223, ciou_loss :226-283), as plain torch expressions (differentiable: the tests take
    autograd gradients of it).  Pinned against the reference's own outputs in tests/golden/ref_box_losses.npz."""
    import math
    lt, rb = torch.max(pred[:, :2], target[:, :2]), torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = ap + ag - overlap + eps
    iou = overlap / union
    e1, e2 = torch.min(pred[:, :2], target[:, :2]), torch.max(pred[:, 2:], target[:, 2:])
    ewh = (e2 - e1).clamp(min=0)
    if kind == 'giou':
        area = ewh[:, 0] * ewh[:, 1] + eps
        return 1 - (iou - (area - union) / area)
    c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + eps
    rho2 = (((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4
            + ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4)
    if kind == 'diou':
        return 1 - (iou - rho2 / c2)
    assert kind == 'ciou'
    w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1] + eps
    w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1] + eps
    v = 4 / math.pi ** 2 * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    return 1 - (iou - (rho2 / c2 + v ** 2 / (1 - iou + v)))


def pointwise_reg_loss(pred, target, kind, beta=1.0):
    """smooth-L1 / L1 / MSE, elementwise (reference lfd/model/losses/smooth_l1_loss.py:11-30, mse_loss.py:11-13): LFD's
    "independent" regression losses.  Differentiable torch expressions; pinned in tests/golden/ref_box_losses.npz."""
    d = (pred - target).abs()
    if kind == 'smooth_l1':
        return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    if kind == 'l1':
        return d
    assert kind == 'mse'
    return (pred - target) ** 2


def bce_with_logits(pred, target):
    """elementwise binary cross-entropy with logits against float targets (reference bce_with_logits_loss.py:28-44 ->
    F.binary_cross_entropy_with_logits, reduction 'none'), written out: max(x,0) - x t + log(1 + exp(-|x|))."""
    return pred.clamp(min=0) - pred * target + torch.log1p(torch.exp(-pred.abs()))


def quality_focal_loss_rows(pred, label, score, beta=2.0):
    """Quality Focal Loss per row (reference gfocal_loss.py:11-52): sum over classes of BCE(x, t) |t - sigmoid(x)|^beta,
    t = score at the row's foreground label, 0 elsewhere (labels outside [0, C) are background)."""
    n, c = pred.shape
    t = torch.zeros_like(pred)
    fg = (label >= 0) & (label < c)
    t[fg, label[fg]] = score[fg].to(pred.dtype)
    return (bce_with_logits(pred, t) * (t - pred.sigmoid()).abs().pow(beta)).sum(1)


def focal_loss_sum(pred, label, gamma=2.0, alpha=0.25):
    """FocalLoss forward via the C restatement (focal_loss.py:39-53), elementwise [N,C]."""
    return torch.from_numpy(c_oracle.sigmoid_focal_loss_fwd(pred.detach().numpy(), label.numpy(), gamma, alpha))


def lfd_loss(arch, cls, reg, sizes, strides, gt_bboxes_list, gt_labels_list):
    """get_loss (lfd.py:284-395) for FocalLoss|CrossEntropyLoss + IoULoss ('sigmoid'/'exp' decode).
    Returns dict(loss, classification_loss, regression_loss, n_pos, n_green) as python floats/ints,
    plus the per-row pieces used by kernel-level parity tests."""
    C = arch['num_classes']
    ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
    cls_t, reg_t, points, rr = lfd_targets(arch, sizes, strides, gt_bboxes_list, gt_labels_list)
    N = cls.shape[0]
    fc = cls.reshape(-1, C + 1 if ce else C)
    fr = reg.reshape(-1, 4)
    ct = cls_t.reshape(-1, C)
    rt = reg_t.reshape(-1, 4)
    green = torch.where(ct.min(dim=-1)[0] >= 0)[0]
    fc, fr, ct, rt = fc[green], fr[green], ct[green], rt[green]
    mx, mi = ct.max(dim=-1)
    pos = torch.where(mx >= 0.001)[0]
    label = mi * (mx >= 0.001) + C * (mx < 0.001)
    if ce:
        cl_el = F.cross_entropy(fc, label, reduction='none')
    elif arch['classification_loss_type'] == 'QualityFocalLoss':      # lfd.py:330-335: targets = [label, max score]
        cl_el = quality_focal_loss_rows(fc, label, mx, 2.0) * 2.0     # TL_LFD_L.py:79-84: beta 2, loss_weight 2
    else:
        cl_el = focal_loss_sum(fc, label)
    cls_loss = cl_el.sum() / (pos.nelement() + 1)
    frp, rtp = fr[pos], rt[pos]
    if pos.nelement() > 0:
        allp = points.repeat(N, 1)[green][pos]
        tx = torch.stack([allp[:, 0] - rtp[:, 0], allp[:, 1] - rtp[:, 1],
                          allp[:, 0] + rtp[:, 2], allp[:, 1] + rtp[:, 3]], -1)
        if arch['distance_to_bbox_mode'] == 'exp':
            d = frp.float().exp()
        else:
            rmax = rr.repeat(N, 1)[green][pos].max(dim=-1)[0]
            d = frp.sigmoid() * rmax[..., None]
        px = torch.stack([allp[:, 0] - d[:, 0], allp[:, 1] - d[:, 1],
                          allp[:, 0] + d[:, 2], allp[:, 1] + d[:, 3]], -1)
        il = torch.from_numpy(c_oracle.iou_loss_fwd(px.numpy(), tx.numpy(), 1e-6))
        reg_loss = il.sum() / pos.nelement()
    else:
        reg_loss = frp.sum()
    return dict(loss=float(cls_loss + reg_loss), classification_loss=float(cls_loss),
                regression_loss=float(reg_loss), n_pos=int(pos.nelement()), n_green=int(green.nelement()),
                cls_targets=cls_t, reg_targets=reg_t, labels=label, green=green, pos=pos)


# ---------------------------------------------------------------- fp16-storage emulation
def _h(t):
    """round-trip through fp16 (the engine's inter-layer storage type)"""
    return t.half().float()


def _fold(sd, conv, norm):
    w = sd[conv + '.weight'].double()
    b = sd[conv + '.bias'].double() if (conv + '.bias') in sd else torch.zeros(w.shape[0], dtype=torch.double)
    if norm is not None:
        s = sd[norm + '.weight'].double() * (sd[norm + '.running_var'].double() + BN_EPS).rsqrt()
        w = w * s.view(-1, 1, 1, 1)
        b = (b - sd[norm + '.running_mean'].double()) * s + sd[norm + '.bias'].double()
    return _h(w.float()), b.float()


def backbone_forward_fp16(sd, arch, x, return_all=False):
    """Backbone half of lfd_forward_fp16: the tapped maps (fp16-valued fp32 tensors).  return_all: additionally the
    list of every fused unit's stored output in execution order (stem convs, then per block: [downsample], conv1, block
    output) -- what the engine keeps in HBM, for per-layer comparisons."""
    pf = '_backbone.'
    seq = {'fast': [(0, 1, 2, 1), (3, 4, 1, 0)], 'faster': [(0, 1, 2, 1), (3, 4, 1, 0), (6, 7, 2, 1), (9, 10, 1, 0)],
           'fastest': [(0, 1, 2, 1), (3, 4, 2, 1)]}[arch['stem_mode']]
    y = _h(x)
    units = []
    for ci, ni, s, p in seq:
        w, b = _fold(sd, f'{pf}_stem.{ci}', f'{pf}_stem.{ni}')
        y = _h(F.relu(F.conv2d(y, w, b, stride=s, padding=p)))
        units.append(('stem%d' % ci, y))
    feats = []
    oi = sorted(tuple(t) for t in arch['out_indices'])
    for i in range(max(s for s, _ in oi) + 1):
        for j in range(arch['body_architecture'][i]):
            bl = f'{pf}stage{i}.{j}.'
            stride = 2 if j == 0 else 1
            ident = y
            if j == 0:
                w, b = _fold(sd, bl + '_downsample.0', bl + '_downsample.1')
                ident = _h(F.conv2d(y, w, b, stride=2))
                units.append(('stage%d.%d.ds' % (i, j), ident))
            nconv = 3 if arch['block_mode'] == 'fast' else 2
            o = y
            for c in range(1, nconv + 1):
                w, b = _fold(sd, bl + f'_conv{c}', bl + f'_norm{c}')
                k = w.shape[-1]
                o = F.conv2d(o, w, b, stride=stride if c == 1 else 1, padding=k // 2)
                if c < nconv:
                    o = _h(F.relu(o))
                    units.append(('stage%d.%d.conv%d' % (i, j, c), o))
            y = _h(F.relu(o + ident))
            units.append(('stage%d.%d' % (i, j), y))
            if (i, j) in oi:
                feats.append(y)
    return (feats, units) if return_all else feats


def head_forward_fp16(sd, arch, feats):
    """Neck + head half of lfd_forward_fp16 from given tapped maps -> (cls [N,P,C'], reg [N,P,4], sizes)."""
    G = arch['gn_groups']
    cls_l, reg_l = [], []
    for i, f in enumerate(feats):
        w, b = _fold(sd, f'_neck.neck{i}.0', f'_neck.neck{i}.1')
        t = _h(F.relu(F.conv2d(f, w, b)))
        hp = f'_head.head{i}_'

        ls = 3 if G else 2

        def tower(name, t):
            for l in range(2):
                yv = F.conv2d(t, _h(sd[f'{name}.{ls * l}.weight']), sd.get(f'{name}.{ls * l}.bias'))
                if G:
                    yv = F.group_norm(yv, G, sd[f'{name}.{ls * l + 1}.weight'], sd[f'{name}.{ls * l + 1}.bias'], GN_EPS)
                t = _h(F.relu(yv))
            return t

        if arch['merge_path_flag']:
            tt = tower(hp + 'merge_path', t)
            c = F.conv2d(tt, _h(sd[hp + 'classification_path.0.weight']), sd[hp + 'classification_path.0.bias'])
            r = F.conv2d(tt, _h(sd[hp + 'regression_path.0.weight']), sd[hp + 'regression_path.0.bias'])
        else:
            c = F.conv2d(tower(hp + 'classification_path', t), _h(sd[hp + f'classification_path.{2 * ls}.weight']),
                         sd[hp + f'classification_path.{2 * ls}.bias'])
            r = F.conv2d(tower(hp + 'regression_path', t), _h(sd[hp + f'regression_path.{2 * ls}.weight']),
                         sd[hp + f'regression_path.{2 * ls}.bias'])
        r = r * sd[f'_head._scales.{i}._scale']
        cls_l.append(c)
        reg_l.append(r)
    sizes = [(c.shape[2], c.shape[3]) for c in cls_l]
    cls = torch.cat([c.permute(0, 2, 3, 1).reshape(c.shape[0], -1, c.shape[1]) for c in cls_l], 1)
    reg = torch.cat([r.permute(0, 2, 3, 1).reshape(r.shape[0], -1, 4) for r in reg_l], 1)
    return cls, reg, sizes


def lfd_forward_fp16(sd, arch, x):
    """Same network as lfd_forward but with the ENGINE's numerics emulated on the CPU: BatchNorm
    folded, weights rounded to fp16, fp32 accumulation, activations rounded to fp16 at every
    fused-unit boundary (after each conv(+residual)+ReLU, the downsample branch, the neck and the
    GroupNorm+ReLU outputs); GroupNorm statistics and the final cls/reg convs stay fp32.
    Gate G2 of SURVEY 8d: the HIP path must match THIS within accumulation-order noise."""
    return head_forward_fp16(sd, arch, backbone_forward_fp16(sd, arch, x))
