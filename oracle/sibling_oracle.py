"""oracle/sibling_oracle.py -- TEST INFRASTRUCTURE.

Plain PyTorch fp32 CPU restatement (functional, driven by a reference-style state_dict) of the sibling
meta-architectures' forward and get_results:

  pyramid_neck_forward   FPN.forward (lfd/model/neck/fpn.py:127-152) and SimpleFPN.forward (simple_fpn.py:141-172)
  fcos_head_forward      FCOSHead.forward (lfd/model/head/fcos_head.py:129-154)
  lfd_head_v1_forward    LFDHeadV1.forward (lfd/model/head/lfd_head.py:322-343)
  fcos_forward           FCOS.forward (lfd/model/fcos.py:414-449)
  lfdv2_forward          LFDv2.forward (lfd/model/lfdv2.py:671-702)
  get_results_single     FCOS._get_results_for_single_image (fcos.py:356-412) and
                         LFDv2._get_results_for_single_image (lfdv2.py:593-669), NMS through lfd_oracle.c

Pinned against the REAL reference modules (imported in the build container) by tests/golden/make_golden_siblings.py ->
tests/golden/ref_sibling_*.npz, checked in tests/test_oracle_golden.py.  Only tests/ may import this module.

`neck` is a plain dict with the reference constructor's kwargs (kind 'FPN' | 'SimpleFPN', num_inputs, num_outputs,
extra_on_input, extra_type, norm_on_lateral ('BatchNorm2d' | ('GroupNorm', groups) | None), relu_on_lateral,
relu_before_extra, neighbouring_mode); `head` likewise (num_layers, norm ('GroupNorm', groups) | 'BatchNorm2d' | None,
num_heads).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle, net_oracle

BN_EPS = net_oracle.BN_EPS
GN_EPS = net_oracle.GN_EPS


def _norm(sd, prefix, x, norm):
    if norm is None:
        return x
    if norm == 'BatchNorm2d':
        return net_oracle._bn(sd, prefix, x)
    kind, groups = norm
    assert kind == 'GroupNorm'
    return F.group_norm(x, groups, sd[prefix + '.weight'], sd[prefix + '.bias'], GN_EPS)


def pyramid_neck_forward(sd, neck, feats, pfx='_neck.'):
    """Lateral 1x1 (+ norm) (+ ReLU); merge by nearest-neighbour upsampling, top-down (fpn.py:132-135) or, in SimpleFPN's
    neighbouring mode, from the finest level up (simple_fpn.py:147-151); then the output paths.  The reference's ReLU in
    front of an extra level is `inplace=True` on the tensor it reads, i.e. it also changes the level it reads FROM
    (fpn.py:66-79 / simple_fpn.py:84-99): reproduced by rewriting that entry."""
    ni, no = neck['num_inputs'], neck['num_outputs']
    lat = []
    for i, f in enumerate(feats):
        y = net_oracle._conv(sd, f'{pfx}lateral{i}.0', f)
        if neck.get('norm_on_lateral'):
            y = _norm(sd, f'{pfx}lateral{i}.1', y, neck['norm_on_lateral'])
        if neck.get('relu_on_lateral'):
            y = F.relu(y)
        lat.append(y)
    if neck['kind'] == 'SimpleFPN' and neck.get('neighbouring_mode'):
        for i in range(ni - 1):
            lat[i] = lat[i] + F.interpolate(lat[i + 1], size=lat[i].shape[2:], mode='nearest')
    else:
        for i in range(ni - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    outs = []
    feats = list(feats)
    for i in range(no):
        if i < ni:
            if neck['kind'] == 'FPN':
                outs.append(net_oracle._conv(sd, f'{pfx}fpn_out{i}.0', lat[i], 1, 1))
            else:
                outs.append(lat[i])
            continue
        from_input = i == ni and neck.get('extra_on_input')
        src = feats[-1] if from_input else outs[-1]
        k = 0
        if neck.get('relu_before_extra'):
            src = F.relu(src)
            if from_input:
                feats[-1] = src
            else:
                outs[-1] = src            # the in-place ReLU reaches the previous output level
            k = 1
        if neck.get('extra_type', 'conv') == 'conv':
            outs.append(net_oracle._conv(sd, f'{pfx}fpn_out{i}.{k}', src, 2, 1))
        else:
            outs.append(F.max_pool2d(src, kernel_size=3, stride=2, padding=1))
    return outs


def fcos_head_forward(sd, head, feats, pfx='_head.'):
    """fcos_head.py:129-154: shared towers, 3x3 output convs, Scale + exp on the regression branch"""
    step = 3 if head.get('norm') else 2

    def tower(name, x):
        for l in range(head['num_layers']):
            x = net_oracle._conv(sd, f'{pfx}{name}.{step * l}', x, 1, 1)
            if head.get('norm'):
                x = _norm(sd, f'{pfx}{name}.{step * l + 1}', x, head['norm'])
            x = F.relu(x)
        return x

    cls, reg, ctr = [], [], []
    for i, f in enumerate(feats):
        tc = tower('_classification_path', f)
        tr = tower('_regression_path', f)
        cls.append(net_oracle._conv(sd, pfx + '_classification', tc, 1, 1))
        ctr.append(net_oracle._conv(sd, pfx + '_centerness', tc, 1, 1))
        r = net_oracle._conv(sd, pfx + '_regression', tr, 1, 1) * sd[f'{pfx}_scales.{i}._scale']
        reg.append(r.float().exp())
    return cls, reg, ctr


def lfd_head_v1_forward(sd, head, feats, pfx='_head.'):
    """LFDHeadV1.forward (lfd/model/head/lfd_head.py:322-343): 1x1 towers (merged or separate, any norm) that end without
    the output convs, then the level's own `_classifiers[i]` / `_regressors[i]`, Scale for the IoU-family losses.
    head: dict(num_conv_layers, norm (None | 'BatchNorm2d' | ('GroupNorm', g)), merge_path_flag, union)"""
    step = 3 if head.get('norm') else 2

    def tower(name, x):
        for l in range(head['num_conv_layers']):
            x = net_oracle._conv(sd, f'{name}.{step * l}', x)
            if head.get('norm'):
                x = _norm(sd, f'{name}.{step * l + 1}', x, head['norm'])
            x = F.relu(x)
        return x

    cls, reg = [], []
    for i, f in enumerate(feats):
        h = f'{pfx}head{i}_'
        if head['merge_path_flag']:
            tc = tr = tower(h + 'merge_path', f)
        else:
            tc, tr = tower(h + 'classification_path', f), tower(h + 'regression_path', f)
        c = net_oracle._conv(sd, f'{pfx}_classifiers.{i}', tc)
        r = net_oracle._conv(sd, f'{pfx}_regressors.{i}', tr)
        if head['union']:
            r = r * sd[f'{pfx}_scales.{i}._scale']
        cls.append(c)
        reg.append(r)
    return cls, reg


def _concat(maps):
    return torch.cat([m.permute(0, 2, 3, 1).reshape(m.shape[0], -1, m.shape[1]) for m in maps], 1)


def fcos_forward(sd, arch, neck, head, x):
    feats = net_oracle.backbone_forward(sd, arch, x)
    cls, reg, ctr = fcos_head_forward(sd, head, pyramid_neck_forward(sd, neck, feats))
    sizes = [(c.shape[2], c.shape[3]) for c in cls]
    return _concat(cls), _concat(reg), _concat(ctr), sizes


def lfdv2_forward(sd, arch, neck, x, head_v1=None):
    """arch: the LFDHead / backbone kwargs of net_oracle; neck None = SimpleNeck; head_v1: lfd_head_v1_forward's dict when
    the head is an LFDHeadV1"""
    feats = net_oracle.backbone_forward(sd, arch, x)
    feats = net_oracle.neck_forward(sd, arch, feats) if neck is None else pyramid_neck_forward(sd, neck, feats)
    cls, reg = net_oracle.head_forward(sd, arch, feats) if head_v1 is None else lfd_head_v1_forward(sd, head_v1, feats)
    sizes = [(c.shape[2], c.shape[3]) for c in cls]
    return _concat(cls), _concat(reg), sizes


def get_results_single(cls_logits, reg, ctr_logits, sizes, strides, ranges, ce_loss, decode, score_thr, iou_thr,
                       pre_nms_limit, post_nms_limit, clamp_hw, resize_scale=1.0, class_agnostic=False):
    """One image.  decode: 'distance' (FCOS: reg already holds distances), 'exp', 'sigmoid' or 'independent'.
    Per level: scores (sigmoid, or softmax without its last column), optional centerness factor, top-k of the points by
    their best final score when the level has more than pre_nms_limit points (ties: lowest point index -- torch.topk
    leaves it open), decode + clamp, then multiclass_nms over the concatenation with score_factors = centerness and
    max_num = post_nms_limit.  Returns dets [k,5], labels [k], kept POINT indices [k] (into the level-concatenated P)."""
    cls_logits = torch.as_tensor(cls_logits, dtype=torch.float32)
    reg = torch.as_tensor(reg, dtype=torch.float32)
    H, W = clamp_hw
    pts = net_oracle.point_coordinates(sizes, strides)
    sc_all, bx_all, f_all, pid_all = [], [], [], []
    p0 = 0
    for l, (h, w) in enumerate(sizes):
        n = h * w
        sl = slice(p0, p0 + n)
        sc = cls_logits[sl].softmax(dim=1)[:, :-1] if ce_loss else cls_logits[sl].sigmoid()
        fac = torch.as_tensor(ctr_logits, dtype=torch.float32).reshape(-1)[sl].sigmoid() if ctr_logits is not None else None
        r = reg[sl]
        p = torch.from_numpy(pts[l])
        idx = torch.arange(n)
        if 0 < pre_nms_limit < n:
            key = (sc * fac[:, None] if fac is not None else sc).max(dim=1)[0].numpy()
            order = np.lexsort((np.arange(n), -key.astype(np.float64)))[:pre_nms_limit]     # descending key, index-stable
            idx = torch.from_numpy(np.sort(order))      # back to point order: only the SET matters (ties aside)
        sc, r, p = sc[idx], r[idx], p[idx]
        if fac is not None:
            fac = fac[idx]
        if decode == 'independent':
            d = r * float(ranges[l][1])
        elif decode == 'exp':
            d = r.float().exp()
        elif decode == 'sigmoid':
            d = r.sigmoid() * float(max(ranges[l]))
        else:
            d = r
        x1 = (p[:, 0] - d[:, 0]).clamp(min=0, max=W)
        y1 = (p[:, 1] - d[:, 1]).clamp(min=0, max=H)
        x2 = (p[:, 0] + d[:, 2]).clamp(min=0, max=W)
        y2 = (p[:, 1] + d[:, 3]).clamp(min=0, max=H)
        bx_all.append(torch.stack([x1, y1, x2, y2], -1))
        sc_all.append(sc)
        if fac is not None:
            f_all.append(fac)
        pid_all.append(idx + p0)
        p0 += n
    sc = torch.cat(sc_all)
    bx = torch.cat(bx_all) / resize_scale
    if f_all:
        sc = sc * torch.cat(f_all)[:, None]          # multiclass_nms: scores * score_factors[:, None] (nms.py:192-193)
    pid = torch.cat(pid_all).numpy()
    dets, labels, cand, _ = c_oracle.multiclass_nms(bx.numpy(), sc.numpy(), score_thr, iou_thr, class_agnostic,
                                                    max_num=post_nms_limit)
    rows = np.argwhere(sc.numpy() > score_thr)[:, 0]     # candidate ordinal -> row of the merged list (nonzero order)
    return dets, labels, (pid[rows[cand]] if len(cand) else cand)
