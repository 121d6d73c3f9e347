"""Thin tensor plumbing over the C ABI (pointers + current HIP stream); no arithmetic here."""
import ctypes as C

import torch

from . import _lib
from ._lib import DetectDesc, F16, F32, check, lib, ptr, require_cuda, stream_ptr

_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only per-device scratch buffer (caller-owned workspace of the C ABI)."""
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError('unsupported dtype %s (float32 / float16 only)' % t.dtype)


# ------------------------------------------------------------------ NMS
def nms_indices(dets, iou_thr):
    """nms_ext.nms: dets [n,5] float32 cuda -> int64 kept indices (score-descending)."""
    require_cuda(dets, 'nms')
    if dets.dim() != 2 or dets.size(1) != 5:
        raise RuntimeError('nms: dets must be [n,5]')
    n = dets.size(0)
    if n == 0:
        return torch.empty(0, dtype=torch.long)   # reference returns an empty CPU long (nms_cuda.cpp:10-11)
    d = dets.contiguous().float()
    with torch.cuda.device(d.device):
        keep = torch.empty(n, dtype=torch.long, device=d.device)
        num = torch.zeros(1, dtype=torch.int32, device=d.device)
        wsb = lib().lfd_nms_workspace_bytes(n)
        ws = _workspace(wsb, d.device)
        check(lib().lfd_nms_f32(ptr(d), n, float(iou_thr), ptr(keep), ptr(num), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_nms_f32')
        k = int(num.item())   # data-dependent output length: the reference path synchronises too
    return keep[:k]


def batched_nms_dets(boxes, scores, labels, iou_thr, class_agnostic=False):
    """batched_nms core: returns (dets[k,5], keep[k] int64)."""
    require_cuda(boxes, 'batched_nms')
    k_in = boxes.size(0)
    if k_in == 0:
        return boxes.new_zeros((0, 5)), torch.empty(0, dtype=torch.long, device=boxes.device)
    b = boxes.contiguous().float()
    s = scores.contiguous().float()
    l = labels.to(device=b.device, dtype=torch.long).contiguous()
    with torch.cuda.device(b.device):
        dets = torch.empty((k_in, 5), dtype=torch.float32, device=b.device)
        keep = torch.empty(k_in, dtype=torch.long, device=b.device)
        num = torch.zeros(1, dtype=torch.int32, device=b.device)
        wsb = lib().lfd_batched_nms_workspace_bytes(k_in)
        ws = _workspace(wsb, b.device)
        check(lib().lfd_batched_nms_f32(ptr(b), ptr(s), ptr(l), k_in, float(iou_thr), int(bool(class_agnostic)),
                                        ptr(dets), ptr(keep), ptr(num), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_batched_nms_f32')
        k = int(num.item())
    return dets[:k], keep[:k]


# ------------------------------------------------------------------ fused decode + threshold + NMS
def make_detect_desc(sizes, strides, ranges, num_classes, num_cls_channels, score_mode, decode_mode,
                     class_agnostic, max_candidates, score_thr, iou_thr):
    if len(sizes) > _lib.MAX_LEVELS:
        raise RuntimeError('at most %d levels' % _lib.MAX_LEVELS)
    d = DetectDesc()
    d.num_levels = len(sizes)
    for i, ((h, w), s, r) in enumerate(zip(sizes, strides, ranges)):
        d.level_h[i], d.level_w[i], d.level_stride[i] = int(h), int(w), int(s)
        d.level_range_lo[i], d.level_range_hi[i] = float(r[0]), float(r[1])
    d.num_classes, d.num_cls_channels = int(num_classes), int(num_cls_channels)
    d.score_mode, d.decode_mode = int(score_mode), int(decode_mode)
    d.class_agnostic, d.max_candidates = int(bool(class_agnostic)), int(max_candidates)
    d.score_thr, d.iou_thr = float(score_thr), float(iou_thr)
    return d


class DetectOutputs(object):
    __slots__ = ('dets', 'labels', 'cand', 'point', 'counts', 'ws')


def detect_outputs(desc, n, dev, out=None):
    """Output tensors + private workspace of a detection step (ops.DetectOutputs); `appendable` marks a workspace whose
    candidate counters have been zeroed for the fused head pass (lfd_head_forward_decode_f16)."""
    cap = desc.max_candidates
    with torch.cuda.device(dev):
        if out is None:
            out = DetectOutputs()
            out.dets = torch.empty((n, cap, 5), dtype=torch.float32, device=dev)
            out.labels = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.cand = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.point = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.counts = torch.empty((n, 4), dtype=torch.int32, device=dev)   # fully written by the kernels
            out.ws = None
        wsb = lib().lfd_detect_workspace_bytes(C.byref(desc), n)
        if getattr(out, 'ws', None) is None or out.ws.numel() < wsb:
            out.ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
    return out


def detect_workspace_reset(desc, n, out):
    """zero the candidate counters of out.ws (once per workspace, before the first lfd_head_forward_decode_f16)"""
    with torch.cuda.device(out.ws.device):
        check(lib().lfd_detect_workspace_reset(C.byref(desc), n, ptr(out.ws), out.ws.numel(), stream_ptr()),
              'lfd_detect_workspace_reset')


def detect_from_candidates(desc, n, out):
    """sort + suppression mask + scan over the candidates a fused head pass appended to out.ws (3 launches)"""
    with torch.cuda.device(out.ws.device):
        check(lib().lfd_detect_from_candidates(C.byref(desc), n, ptr(out.dets), ptr(out.labels), ptr(out.cand), ptr(out.point),
                                               ptr(out.counts), ptr(out.ws), out.ws.numel(), stream_ptr()),
              'lfd_detect_from_candidates')
    return out


def detect_batched(desc, cls, reg, meta, out=None):
    """Enqueues the whole post-processing of a batch; returns device-resident outputs (no sync)."""
    require_cuda(cls, 'detect')
    n = cls.size(0)
    cls = cls.contiguous()
    reg = reg.contiguous()
    if cls.dtype != reg.dtype:
        raise RuntimeError('cls / reg dtype mismatch')
    dev = cls.device
    cap = desc.max_candidates
    with torch.cuda.device(dev):
        if out is None:
            out = DetectOutputs()
            out.dets = torch.empty((n, cap, 5), dtype=torch.float32, device=dev)
            out.labels = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.cand = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.point = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.counts = torch.empty((n, 4), dtype=torch.int32, device=dev)   # fully written by k_count/k_scatter/k_scan
            out.ws = None
        wsb = lib().lfd_detect_workspace_bytes(C.byref(desc), n)
        # the workspace belongs to the output set (not the process-wide scratch): passes of different output sets may run
        # concurrently on different streams
        if getattr(out, 'ws', None) is None or out.ws.numel() < wsb:
            out.ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
        ws = out.ws
        check(lib().lfd_detect_batched(C.byref(desc), n, ptr(cls), ptr(reg), _dtype_code(cls), ptr(meta),
                                       ptr(out.dets), ptr(out.labels), ptr(out.cand), ptr(out.point),
                                       ptr(out.counts), ptr(ws), ws.numel(), stream_ptr()), 'lfd_detect_batched')
    return out


def detect_batched_ex(desc, cls, reg, meta, centerness=None, pre_nms_limit=-1, post_nms_limit=-1, out=None):
    """detect_batched for the sibling meta-architectures (FCOS / LFDv2 get_results): optional centerness logits [N,P]
    (score factor sigmoid(centerness)), per-level pre-NMS top-k, post-NMS cap (lfd_detect_batched_ex)."""
    require_cuda(cls, 'detect')
    n = cls.size(0)
    cls = cls.contiguous()
    reg = reg.contiguous()
    if cls.dtype != reg.dtype:
        raise RuntimeError('cls / reg dtype mismatch')
    if centerness is not None:
        centerness = centerness.contiguous()
        if centerness.dtype != cls.dtype or centerness.numel() != n * cls.size(1):
            raise RuntimeError('centerness must be [N,P] (or [N,P,1]) in the dtype of cls')
    dev = cls.device
    cap = desc.max_candidates
    ext = _lib.DetectExt(int(pre_nms_limit), int(post_nms_limit))
    with torch.cuda.device(dev):
        if out is None:
            out = DetectOutputs()
            out.dets = torch.empty((n, cap, 5), dtype=torch.float32, device=dev)
            out.labels = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.cand = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.point = torch.empty((n, cap), dtype=torch.int32, device=dev)
            out.counts = torch.empty((n, 4), dtype=torch.int32, device=dev)
            out.ws = None
        wsb = lib().lfd_detect_ex_workspace_bytes(C.byref(desc), n)
        if getattr(out, 'ws', None) is None or out.ws.numel() < wsb:
            out.ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
        ws = out.ws
        check(lib().lfd_detect_batched_ex(C.byref(desc), C.byref(ext), n, ptr(cls), ptr(reg), ptr(centerness),
                                          _dtype_code(cls), ptr(meta), ptr(out.dets), ptr(out.labels), ptr(out.cand),
                                          ptr(out.point), ptr(out.counts), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_detect_batched_ex')
    return out


def decode_all(desc, cls, reg, meta):
    require_cuda(cls, 'decode')
    n, p = cls.size(0), cls.size(1)
    cls = cls.contiguous()
    reg = reg.contiguous()
    with torch.cuda.device(cls.device):
        boxes = torch.empty((n, p, 4), dtype=torch.float32, device=cls.device)
        scores = torch.empty((n, p, desc.num_classes), dtype=torch.float32, device=cls.device)
        check(lib().lfd_decode_all(C.byref(desc), n, ptr(cls), ptr(reg), _dtype_code(cls), ptr(meta), ptr(boxes),
                                   ptr(scores), stream_ptr()), 'lfd_decode_all')
    return boxes, scores


# ------------------------------------------------------------------ losses
def focal_forward(logits, targets, gamma, alpha):
    require_cuda(logits, 'sigmoid_focal_loss forward')
    if logits.dim() != 2:
        raise RuntimeError('logits should be NxClass')
    x = logits.contiguous()
    t = targets.contiguous()
    if t.dtype != torch.long:
        raise RuntimeError('targets must be int64')
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().lfd_sigmoid_focal_loss_fwd(ptr(x), ptr(t), x.size(0), x.size(1), float(gamma), float(alpha),
                                               ptr(out), _dtype_code(x), stream_ptr()), 'lfd_sigmoid_focal_loss_fwd')
    return out


def focal_backward(logits, targets, d_losses, gamma, alpha):
    require_cuda(logits, 'sigmoid_focal_loss backward')
    x = logits.contiguous()
    t = targets.contiguous()
    g = d_losses.contiguous().to(x.dtype)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().lfd_sigmoid_focal_loss_bwd(ptr(x), ptr(t), ptr(g), x.size(0), x.size(1), float(gamma),
                                               float(alpha), ptr(out), _dtype_code(x), stream_ptr()),
              'lfd_sigmoid_focal_loss_bwd')
    return out


def focal_sum(logits, targets, gamma, alpha):
    require_cuda(logits, 'sigmoid_focal_loss sum')
    x = logits.contiguous().float()
    t = targets.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws = _workspace(lib().lfd_reduce_workspace_bytes(), x.device)
        check(lib().lfd_sigmoid_focal_loss_sum_f32(ptr(x), ptr(t), x.size(0), x.size(1), float(gamma), float(alpha),
                                                   ptr(out), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_sigmoid_focal_loss_sum_f32')
    return out


def iou_loss_forward(pred, target, eps):
    require_cuda(pred, 'iou_loss forward')
    a = pred.contiguous().float()
    b = target.contiguous().float()
    out = torch.empty(a.size(0), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib().lfd_iou_loss_fwd_f32(ptr(a), ptr(b), a.size(0), float(eps), ptr(out), stream_ptr()),
              'lfd_iou_loss_fwd_f32')
    return out


def iou_loss_backward(pred, target, d_loss, eps):
    require_cuda(pred, 'iou_loss backward')
    a = pred.contiguous().float()
    b = target.contiguous().float()
    g = d_loss.contiguous().float()
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        check(lib().lfd_iou_loss_bwd_f32(ptr(a), ptr(b), ptr(g), a.size(0), float(eps), ptr(out), stream_ptr()),
              'lfd_iou_loss_bwd_f32')
    return out


BOX_LOSS_KINDS = {'giou': 1, 'diou': 2, 'ciou': 3}


def box_loss(pred, target, kind, eps, want_grad=True):
    """GIoU / DIoU / CIoU loss of aligned xyxy box pairs [n,4] -> (loss [n], d loss / d pred [n,4] or None), one launch
    (lfd_box_loss_f32: the gradient comes from forward-mode differentiation inside the kernel)."""
    require_cuda(pred, 'box_loss')
    p = pred.detach().contiguous().float()
    t = target.detach().contiguous().float()
    if p.shape != t.shape or p.dim() != 2 or p.size(1) != 4:
        raise ValueError('box_loss: pred / target must both be [n, 4]')
    n = p.size(0)
    loss = torch.empty(n, dtype=torch.float32, device=p.device)
    grad = torch.empty_like(p) if want_grad else None
    with torch.cuda.device(p.device):
        check(lib().lfd_box_loss_f32(ptr(p), ptr(t), n, BOX_LOSS_KINDS[kind], float(eps), ptr(loss), ptr(grad), stream_ptr()),
              'lfd_box_loss_f32')
    return loss, grad


def bce_with_logits(logits, targets, want_grad=True):
    """elementwise BCE-with-logits against float targets -> (loss, d loss / d logits or None), same shape"""
    require_cuda(logits, 'bce_with_logits')
    x = logits.detach().contiguous().float()
    t = targets.detach().contiguous().float()
    if x.shape != t.shape:
        raise ValueError('bce_with_logits: logits / targets shapes differ')
    loss = torch.empty_like(x)
    grad = torch.empty_like(x) if want_grad else None
    with torch.cuda.device(x.device):
        check(lib().lfd_bce_with_logits_f32(ptr(x), ptr(t), x.numel(), ptr(loss), ptr(grad), stream_ptr()),
              'lfd_bce_with_logits_f32')
    return loss, grad


def quality_focal_loss(logits, labels, scores, beta, want_grad=True):
    """QFL: logits [N,C], labels [N] int64 (label outside [0, C) = background), scores [N] -> (loss [N], grad [N,C])"""
    require_cuda(logits, 'quality_focal_loss')
    x = logits.detach().contiguous().float()
    lab = labels.detach().contiguous().long()
    sc = scores.detach().contiguous().float()
    if x.dim() != 2 or lab.shape != (x.size(0),) or sc.shape != (x.size(0),):
        raise ValueError('quality_focal_loss: logits [N,C], labels [N], scores [N] expected')
    loss = torch.empty(x.size(0), dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x) if want_grad else None
    with torch.cuda.device(x.device):
        check(lib().lfd_quality_focal_loss_f32(ptr(x), ptr(lab), ptr(sc), x.size(0), x.size(1), float(beta), ptr(loss),
                                               ptr(grad), stream_ptr()), 'lfd_quality_focal_loss_f32')
    return loss, grad


POINTWISE_LOSS_KINDS = {'smooth_l1': 1, 'l1': 2, 'mse': 3}


def pointwise_loss(pred, target, kind, beta=1.0, want_grad=True):
    """smooth-L1 / L1 / MSE of same-shaped tensors -> (loss, d loss / d pred or None), same shape, one launch."""
    require_cuda(pred, 'pointwise_loss')
    p = pred.detach().contiguous().float()
    t = target.detach().contiguous().float()
    if p.shape != t.shape:
        raise ValueError('pointwise_loss: pred / target shapes differ')
    loss = torch.empty_like(p)
    grad = torch.empty_like(p) if want_grad else None
    with torch.cuda.device(p.device):
        check(lib().lfd_pointwise_loss_f32(ptr(p), ptr(t), p.numel(), POINTWISE_LOSS_KINDS[kind], float(beta), ptr(loss),
                                           ptr(grad), stream_ptr()), 'lfd_pointwise_loss_f32')
    return loss, grad


def cross_entropy_forward(logits, labels):
    """F.cross_entropy(logits, labels, reduction='none') on the device (cross_entropy_loss.py:12-16)."""
    require_cuda(logits, 'cross_entropy forward')
    x = logits.contiguous().float()
    t = labels.contiguous().long()
    out = torch.empty(x.size(0), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().lfd_cross_entropy_fwd_f32(ptr(x), ptr(t), x.size(0), x.size(1), ptr(out), stream_ptr()),
              'lfd_cross_entropy_fwd_f32')
    return out


def cross_entropy_backward(logits, labels, d_loss):
    require_cuda(logits, 'cross_entropy backward')
    x = logits.contiguous().float()
    t = labels.contiguous().long()
    g = d_loss.contiguous().float()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().lfd_cross_entropy_bwd_f32(ptr(x), ptr(t), ptr(g), x.size(0), x.size(1), ptr(out), stream_ptr()),
              'lfd_cross_entropy_bwd_f32')
    return out


ASSIGN_MODES = {'longer': 0, 'shorter': 1, 'sqrt': 2, 'dist': 3}


def make_assign_desc(n, sizes, strides, reg_ranges, gray_ranges, num_classes, assign_mode, independent):
    """-> (lfd_assign_desc_t, total points per image)"""
    d = _lib.AssignDesc()
    d.n, d.num_levels = int(n), len(sizes)
    total = 0
    for i, (h, w) in enumerate(sizes):
        d.level_h[i], d.level_w[i], d.stride[i] = int(h), int(w), int(strides[i])
        d.reg_lo[i], d.reg_hi[i] = int(reg_ranges[i][0]), int(reg_ranges[i][1])
        d.gray_lo[i], d.gray_hi[i] = int(gray_ranges[i][0]), int(gray_ranges[i][1])
        total += int(h) * int(w)
    d.total_points, d.num_classes = total, int(num_classes)
    d.assign_mode, d.independent = ASSIGN_MODES[assign_mode], int(bool(independent))
    return d, total


def assign_targets_device(desc, total, num_classes, boxes, labels, offs):
    """targets from device-resident annotations: boxes [K,4] fp32 xywh, labels [K] int64, offs [N+1] int32 (image i owns
    boxes offs[i] .. offs[i+1]; K may exceed offs[N]: capacity of a static buffer) -- what a captured training iteration
    (lfd_amd.train.GraphedTrainStep) launches"""
    require_cuda(boxes, 'assign_targets')
    return _assign_launch(desc, desc.n, total, num_classes, boxes, labels, offs, boxes.device)


def assign_targets(sizes, strides, reg_ranges, gray_ranges, num_classes, assign_mode, independent, gt_bboxes_list,
                   gt_labels_list):
    """LFD.annotation_to_target (lfd.py:109-259) for a batch: -> cls targets [N,P,C], reg targets [N,P,4] (fp32,
    device resident).  gt_bboxes_list[i]: [G_i,4] xywh cuda tensor, gt_labels_list[i]: [G_i] int64."""
    n = len(gt_bboxes_list)
    dev = gt_bboxes_list[0].device
    require_cuda(gt_bboxes_list[0], 'assign_targets')
    d, total = make_assign_desc(n, sizes, strides, reg_ranges, gray_ranges, num_classes, assign_mode, independent)
    counts = [int(b.size(0)) for b in gt_bboxes_list]
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32).to(dev)
    if sum(counts):
        boxes = torch.cat([b.reshape(-1, 4) for b in gt_bboxes_list], 0).contiguous().float()
        labels = torch.cat([l.reshape(-1) for l in gt_labels_list], 0).contiguous().long()
    else:
        boxes = torch.zeros((1, 4), dtype=torch.float32, device=dev)
        labels = torch.zeros((1,), dtype=torch.int64, device=dev)
    return _assign_launch(d, n, total, num_classes, boxes, labels, offs, dev)


def assign_targets_from_host(sizes, strides, reg_ranges, gray_ranges, num_classes, assign_mode, independent,
                             annotation_batch, device):
    """Same as assign_targets for the annotation_batch LFD.get_loss receives (list of (bboxes numpy [G,4] xywh, labels
    numpy [G]) per image, lfd.py:290-297): concatenated on the host and uploaded with three copies instead of two per image."""
    import numpy as np
    n = len(annotation_batch)
    d, total = make_assign_desc(n, sizes, strides, reg_ranges, gray_ranges, num_classes, assign_mode, independent)
    bl = [np.asarray(b, dtype=np.float32).reshape(-1, 4) for b, _ in annotation_batch]
    ll = [np.asarray(l, dtype=np.int64).reshape(-1) for _, l in annotation_batch]
    counts = [b.shape[0] for b in bl]
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(device)
    if sum(counts):
        boxes = torch.from_numpy(np.ascontiguousarray(np.concatenate(bl, 0))).to(device)
        labels = torch.from_numpy(np.ascontiguousarray(np.concatenate(ll, 0))).to(device)
    else:
        boxes = torch.zeros((1, 4), dtype=torch.float32, device=device)
        labels = torch.zeros((1,), dtype=torch.int64, device=device)
    return _assign_launch(d, n, total, num_classes, boxes, labels, offs, device)


def _assign_launch(d, n, total, num_classes, boxes, labels, offs, dev):
    cls_t = torch.empty((n, total, int(num_classes)), dtype=torch.float32, device=dev)
    reg_t = torch.empty((n, total, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().lfd_assign_targets_f32(C.byref(d), ptr(boxes), ptr(labels), ptr(offs), ptr(cls_t), ptr(reg_t), stream_ptr()),
              'lfd_assign_targets_f32')
    return cls_t, reg_t


def make_loss_desc(n, sizes, strides, reg_ranges, num_classes, cross_entropy, decode_mode, gamma=2.0, alpha=0.25,
                   iou_eps=1e-6, cls_loss_weight=1.0, reg_loss_weight=1.0, cls_weighted=False, reg_weighted=False):
    """lfd_loss_desc_t for the fused get_loss kernels (decode_mode: 'sigmoid' | 'exp', lfd.py:366-374)."""
    d = _lib.LossDesc()
    d.n, d.num_levels = int(n), len(sizes)
    total = 0
    for i, (h, w) in enumerate(sizes):
        d.level_h[i], d.level_w[i], d.stride[i] = int(h), int(w), int(strides[i])
        d.range_max[i] = float(max(reg_ranges[i]))
        total += int(h) * int(w)
    d.total_points, d.num_classes = total, int(num_classes)
    d.cls_loss = 1 if cross_entropy else 0
    d.decode_mode = {'sigmoid': 0, 'exp': 1}[decode_mode]
    d.gamma, d.alpha, d.iou_eps = float(gamma), float(alpha), float(iou_eps)
    d.cls_loss_weight, d.reg_loss_weight = float(cls_loss_weight), float(reg_loss_weight)
    d.cls_weighted, d.reg_weighted = int(bool(cls_weighted)), int(bool(reg_weighted))
    return d


def _loss_inputs(desc, pred_cls, pred_reg, cls_t, reg_t):
    require_cuda(pred_cls, 'get_loss')
    ch = desc.num_classes + (1 if desc.cls_loss else 0)
    rows = desc.n * desc.total_points
    ts = [pred_cls.detach().contiguous().float(), pred_reg.detach().contiguous().float(),
          cls_t.contiguous().float(), reg_t.contiguous().float()]
    for t, c in zip(ts, (ch, 4, desc.num_classes, 4)):
        if t.numel() != rows * c:
            raise ValueError('get_loss: tensor with %d elements, expected %d rows x %d' % (t.numel(), rows, c))
    return ts


def get_loss_sums(desc, pred_cls, pred_reg, cls_t, reg_t):
    """First half of the fused get_loss forward: this rank's float64[8] sums (lfd_get_loss_sums_f32).  Under image-parallel
    training they are all-reduced before `get_loss_finalize` (lfd.py:340,383: `n_pos + 1` / `n_pos` are global-batch counts)."""
    pc, pr, ct, rt = _loss_inputs(desc, pred_cls, pred_reg, cls_t, reg_t)
    dev = pc.device
    nbytes = lib().lfd_get_loss_workspace_bytes()
    ws = _workspace(nbytes, dev)
    sums = torch.empty(8, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib().lfd_get_loss_sums_f32(C.byref(desc), ptr(pc), ptr(pr), ptr(ct), ptr(rt), ptr(ws), nbytes, ptr(sums),
                                          stream_ptr()), 'lfd_get_loss_sums_f32')
    return sums


def get_loss_finalize(desc, sums, gsums, rank_scale=1.0):
    """Second half: local + global sums -> float32[8] {classification_loss, regression_loss, loss, n_pos, avg_cls, avg_reg,
    n_green, rank_scale} (lfd_get_loss_finalize_f32)"""
    out = torch.empty(8, dtype=torch.float32, device=sums.device)
    with torch.cuda.device(sums.device):
        check(lib().lfd_get_loss_finalize_f32(C.byref(desc), ptr(sums), ptr(gsums), float(rank_scale), ptr(out),
                                              stream_ptr()), 'lfd_get_loss_finalize_f32')
    return out


def get_loss_forward(desc, pred_cls, pred_reg, cls_t, reg_t, reduce_sums=None, rank_scale=1.0):
    """Fused LFD.get_loss forward (lfd.py:284-395) -> float32[8] device tensor
    {classification_loss, regression_loss, loss, n_pos, avg_cls, avg_reg, n_green, rank_scale}.
    reduce_sums: optional callable mapping the local float64[8] sums tensor to the global one (the
    all-reduce over image-parallel ranks, lfd_amd.parallel.global_count)."""
    sums = get_loss_sums(desc, pred_cls, pred_reg, cls_t, reg_t)
    gsums = reduce_sums(sums) if reduce_sums is not None else sums
    return get_loss_finalize(desc, sums, gsums, rank_scale)


def get_loss_backward(desc, pred_cls, pred_reg, cls_t, reg_t, finalized, grad_out):
    """-> (d pred_cls, d pred_reg), dense, for grad_out[3] = d/d{classification_loss, regression_loss, loss}."""
    pc, pr, ct, rt = _loss_inputs(desc, pred_cls, pred_reg, cls_t, reg_t)
    g = grad_out.contiguous().float()
    gc, gr = torch.empty_like(pc), torch.empty_like(pr)
    with torch.cuda.device(pc.device):
        check(lib().lfd_get_loss_bwd_f32(C.byref(desc), ptr(pc), ptr(pr), ptr(ct), ptr(rt), ptr(finalized), ptr(g),
                                         ptr(gc), ptr(gr), stream_ptr()), 'lfd_get_loss_bwd_f32')
    return gc, gr


# ------------------------------------------------------------------ conv stack (NHWC fp16)
def pack_conv_weight(w):
    """[Cout,Cin,k,k] float (BN already folded) -> MFMA fragment order [Cout/32][k*k*Cin/16][64][8] fp16.
    lane = 32*half + cout_local; the 8 halfs of a lane are channels 16q + 8*half + 0..7 of tap (r,s),
    k-step index = (r*k + s)*(Cin/16) + q  (csrc/conv.hip)."""
    cout, cin, ks, _ = w.shape
    assert cout % 32 == 0 and cin % 16 == 0
    nq = cin // 16
    w5 = w.detach().float().permute(0, 2, 3, 1).reshape(cout // 32, 32, ks, ks, nq, 2, 8)
    wp = w5.permute(0, 2, 3, 4, 5, 1, 6).reshape(cout // 32, ks * ks * nq, 64, 8)
    return wp.half().contiguous()


_zeros = {}


def zero_line(device):
    key = (device.type, device.index)
    z = _zeros.get(key)
    if z is None:
        z = torch.zeros(4096, dtype=torch.uint8, device=device)
        _zeros[key] = z
    return z


def conv2d_nhwc(x, w_packed, bias, cin, cout, ks, stride, relu, residual=None, tail=None, out=None):
    """x [N,H,W,cin] fp16 -> [N,OH,OW,cout] fp16.  tail = (w2_packed, bias2, relu2) chains a 1x1."""
    require_cuda(x, 'conv2d')
    if x.dtype != torch.float16 or not x.is_contiguous():
        raise RuntimeError('conv2d_nhwc: x must be contiguous fp16 NHWC')
    n, h, w_, c = x.shape
    if c != cin:
        raise RuntimeError('conv2d_nhwc: channel mismatch')
    pad = ks // 2
    oh = (h + 2 * pad - ks) // stride + 1
    ow = (w_ + 2 * pad - ks) // stride + 1
    d = _lib.ConvDesc(n, h, w_, cin, cout, ks, stride, int(relu), cout if tail else 0,
                      int(bool(tail[2])) if tail else 0)
    with torch.cuda.device(x.device):
        if out is None:
            out = torch.empty((n, oh, ow, cout), dtype=torch.float16, device=x.device)
        check(lib().lfd_conv2d_nhwc_f16(C.byref(d), ptr(x), ptr(out), ptr(w_packed), ptr(bias), ptr(residual),
                                        ptr(tail[0]) if tail else None, ptr(tail[1]) if tail else None,
                                        ptr(zero_line(x.device)), stream_ptr()), 'lfd_conv2d_nhwc_f16')
    return out


def conv2d_nhwc_f32out(x, w_packed, bias, cin, cout, ks, stride):
    """x [N,H,W,cin] fp16 -> conv + bias as fp32 [N,OH,OW,cout] (the MFMA accumulators, un-rounded): the output convs of
    the sibling heads, whose logits leave the network as fp32 (lfd_conv2d_nhwc_f16_acc32)."""
    require_cuda(x, 'conv2d')
    if x.dtype != torch.float16 or not x.is_contiguous() or x.shape[3] != cin:
        raise RuntimeError('conv2d_nhwc_f32out: x must be contiguous fp16 NHWC with %d channels' % cin)
    n, h, w_, _ = x.shape
    pad = ks // 2
    oh = (h + 2 * pad - ks) // stride + 1
    ow = (w_ + 2 * pad - ks) // stride + 1
    d = _lib.ConvDesc(n, h, w_, cin, cout, ks, stride, 0, 0, 0)
    with torch.cuda.device(x.device):
        out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
        check(lib().lfd_conv2d_nhwc_f16_acc32(C.byref(d), ptr(x), ptr(out), ptr(w_packed), ptr(bias),
                                              ptr(zero_line(x.device)), stream_ptr()), 'lfd_conv2d_nhwc_f16_acc32')
    return out


def upsample_nearest_add_(dst, src):
    """dst [N,H,W,C] += nearest-neighbour resize of src [N,h,w,C] (fp16 NHWC, in place): the FPN merge step"""
    _nhwc16(dst, 'upsample_nearest_add')
    _nhwc16(src, 'upsample_nearest_add')
    n, H, W, c = dst.shape
    if src.shape[0] != n or src.shape[3] != c:
        raise RuntimeError('upsample_nearest_add: batch / channel mismatch')
    with torch.cuda.device(dst.device):
        check(lib().lfd_upsample_nearest_add_nhwc_f16(ptr(dst), ptr(src), n, H, W, src.shape[1], src.shape[2], c, stream_ptr()),
              'lfd_upsample_nearest_add_nhwc_f16')
    return dst


def relu_(x):
    _nhwc16(x, 'relu_')
    with torch.cuda.device(x.device):
        check(lib().lfd_relu_inplace_f16(ptr(x), x.numel(), stream_ptr()), 'lfd_relu_inplace_f16')
    return x


def maxpool3x3s2(x):
    _nhwc16(x, 'maxpool3x3s2')
    n, h, w_, c = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty((n, (h - 1) // 2 + 1, (w_ - 1) // 2 + 1, c), dtype=torch.float16, device=x.device)
        check(lib().lfd_maxpool3x3s2_nhwc_f16(ptr(x), ptr(out), n, h, w_, c, stream_ptr()), 'lfd_maxpool3x3s2_nhwc_f16')
    return out


def pack_level_outputs(src, dst, c0, count, point_offset, scale=1.0, exp=False):
    """channels c0..c0+count of src [N,h,w,cs] fp32 -> dst[:, point_offset:point_offset+h*w, :count] (dst [N,P,count] fp32),
    times `scale`, optionally through expf"""
    require_cuda(src, 'pack_level_outputs')
    n, h, w_, cs = src.shape
    if dst.dtype != torch.float32 or src.dtype != torch.float32 or not dst.is_contiguous() or not src.is_contiguous():
        raise RuntimeError('pack_level_outputs: contiguous fp32 tensors expected')
    if dst.size(0) != n or dst.size(2) != count:
        raise RuntimeError('pack_level_outputs: shape mismatch')
    with torch.cuda.device(src.device):
        check(lib().lfd_pack_level_outputs_f32(ptr(src), ptr(dst), n, h * w_, cs, c0, count, dst.size(1), point_offset,
                                               float(scale), int(bool(exp)), stream_ptr()), 'lfd_pack_level_outputs_f32')
    return dst


def fasterblock_fused(x, w1_packed, b1, w2_packed, b2, out=None):
    """relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x) for NHWC fp16 [N,H,W,64] in one launch (csrc/block.hip)."""
    require_cuda(x, 'fasterblock_fused')
    if x.dtype != torch.float16 or not x.is_contiguous() or x.dim() != 4 or x.shape[3] != 64:
        raise RuntimeError('fasterblock_fused: x must be contiguous fp16 NHWC with 64 channels')
    n, h, w_, _ = x.shape
    with torch.cuda.device(x.device):
        if out is None:
            out = torch.empty_like(x)
        check(lib().lfd_fasterblock_fused_f16(n, h, w_, ptr(x), ptr(out), ptr(w1_packed), ptr(b1), ptr(w2_packed), ptr(b2),
                                              ptr(zero_line(x.device)), stream_ptr()), 'lfd_fasterblock_fused_f16')
    return out


def fasterblock128_fused(x, w1_packed, b1, w2_packed, b2, out=None):
    """relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x) for NHWC fp16 [N,H,W,128] on the small maps of the last backbone
    stage in one launch (csrc/block128.hip); bit-identical to two conv2d_nhwc launches on their split-K path."""
    require_cuda(x, 'fasterblock128_fused')
    if x.dtype != torch.float16 or not x.is_contiguous() or x.dim() != 4 or x.shape[3] != 128:
        raise RuntimeError('fasterblock128_fused: x must be contiguous fp16 NHWC with 128 channels')
    n, h, w_, _ = x.shape
    with torch.cuda.device(x.device):
        if out is None:
            out = torch.empty_like(x)
        check(lib().lfd_fasterblock128_fused_f16(n, h, w_, ptr(x), ptr(out), ptr(w1_packed), ptr(b1), ptr(w2_packed), ptr(b2),
                                                 stream_ptr()), 'lfd_fasterblock128_fused_f16')
    return out


def conv2d_downsample_nhwc(x, w_packed, bias, wd_packed, bd, relu=True):
    """(relu(conv3x3 stride 2 (x) + bias), conv1x1 stride 2 (x) + bd) for NHWC fp16 [N,H,W,64]: the first launch of a
    stage's first block with its identity branch riding on the centre tap (lfd_conv2d_downsample_nhwc_f16)."""
    require_cuda(x, 'conv2d_downsample')
    if x.dtype != torch.float16 or not x.is_contiguous() or x.dim() != 4:
        raise RuntimeError('conv2d_downsample_nhwc: x must be contiguous fp16 NHWC')
    n, h, w_, c = x.shape
    d = _lib.ConvDesc(n, h, w_, c, c, 3, 2, int(relu), 0, 0)
    with torch.cuda.device(x.device):
        out = torch.empty((n, (h - 1) // 2 + 1, (w_ - 1) // 2 + 1, c), dtype=torch.float16, device=x.device)
        ident = torch.empty_like(out)
        check(lib().lfd_conv2d_downsample_nhwc_f16(C.byref(d), ptr(x), ptr(out), ptr(w_packed), ptr(bias), ptr(wd_packed), ptr(bd),
                                                   ptr(ident), ptr(zero_line(x.device)), stream_ptr()),
              'lfd_conv2d_downsample_nhwc_f16')
    return out, ident


def downblock_fused(x, w1_packed, b1, wd_packed, bd, w2_packed, b2, out=None):
    """relu(conv3x3(relu(conv3x3_s2(x, w1) + b1), w2) + b2 + conv1x1_s2(x, wd) + bd) for NHWC fp16 [N,H,W,64] in one launch
    (csrc/down.hip): the first block of a backbone stage."""
    require_cuda(x, 'downblock_fused')
    if x.dtype != torch.float16 or not x.is_contiguous() or x.dim() != 4 or x.shape[3] != 64:
        raise RuntimeError('downblock_fused: x must be contiguous fp16 NHWC with 64 channels')
    n, h, w_, _ = x.shape
    with torch.cuda.device(x.device):
        if out is None:
            out = torch.empty((n, (h - 1) // 2 + 1, (w_ - 1) // 2 + 1, 64), dtype=torch.float16, device=x.device)
        check(lib().lfd_downblock_fused_f16(n, h, w_, ptr(x), ptr(out), ptr(w1_packed), ptr(b1), ptr(wd_packed), ptr(bd),
                                            ptr(w2_packed), ptr(b2), ptr(zero_line(x.device)), stream_ptr()),
              'lfd_downblock_fused_f16')
    return out


# ------------------------------------------------------------------ training-mode conv stack (csrc/train.hip)
_train_ws = {}


def train_workspace(device):
    """scratch of the training kernels (per-block partial sums), one per (device, stream): launch chains on different streams
    -- the pyramid levels and the weight gradients of lfd_amd.train_engine's schedule -- must not share it"""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _train_ws.get(key)
    if ws is None:
        ws = torch.empty(lib().lfd_train_workspace_bytes(), dtype=torch.uint8, device=device)
        _train_ws[key] = ws
    return ws


def _nhwc16(t, what):
    require_cuda(t, what)
    if t.dtype != torch.float16 or not t.is_contiguous() or t.dim() != 4:
        raise RuntimeError('%s: contiguous fp16 NHWC tensor expected' % what)
    return t


def bn_train_stats(y, eps, momentum, running_mean=None, running_var=None):
    """Batch statistics of y [N,H,W,C] -> float32[2*C] (mean, rstd); updates the running statistics in place."""
    _nhwc16(y, 'bn_train_stats')
    c = y.size(3)
    stats = torch.empty(2 * c, dtype=torch.float32, device=y.device)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_bn_train_stats_f16(ptr(y), y.numel() // c, c, float(eps), float(momentum), ptr(running_mean),
                                           ptr(running_var), ptr(ws), ws.numel(), ptr(stats), stream_ptr()),
              'lfd_bn_train_stats_f16')
    return stats


def conv2d_bn_stats(x, w_packed, bias, cin, cout, ks, stride, eps, momentum, running_mean=None, running_var=None):
    """The conv of a train-mode Conv2d(bias=False) -> BatchNorm2d unit with the batch statistics of its output taken in the
    conv's own epilogue: x [N,H,W,cin] fp16 -> (y [N,OH,OW,cout] fp16, float32[2*cout] (mean, rstd)); updates the running
    statistics in place (lfd_conv2d_bn_stats_nhwc_f16)."""
    _nhwc16(x, 'conv2d_bn_stats')
    n, h, w_, c = x.shape
    if c != cin:
        raise RuntimeError('conv2d_bn_stats: channel mismatch')
    pad = ks // 2
    oh = (h + 2 * pad - ks) // stride + 1
    ow = (w_ + 2 * pad - ks) // stride + 1
    d = _lib.ConvDesc(n, h, w_, cin, cout, ks, stride, 0, 0, 0)
    ws = train_workspace(x.device)
    with torch.cuda.device(x.device):
        y = torch.empty((n, oh, ow, cout), dtype=torch.float16, device=x.device)
        stats = torch.empty(2 * cout, dtype=torch.float32, device=x.device)
        check(lib().lfd_conv2d_bn_stats_nhwc_f16(C.byref(d), ptr(x), ptr(y), ptr(w_packed), ptr(bias), ptr(zero_line(x.device)),
                                                 float(eps), float(momentum), ptr(running_mean), ptr(running_var), ptr(ws),
                                                 ws.numel(), ptr(stats), stream_ptr()), 'lfd_conv2d_bn_stats_nhwc_f16')
    return y, stats


def conv1x1_of_bn_relu_bn_stats(y_in, in_stats, in_gamma, in_beta, w_packed, bias, cout, eps, momentum, running_mean=None,
                                running_var=None):
    """conv2d_bn_stats for a 1x1 stride-1 conv whose input is relu(BatchNorm(y_in)) of a unit WITHOUT a residual: the
    normalisation + ReLU happens on the activation fragments inside the conv kernel (bit-identical to bn_train_apply first),
    the producer's z tensor is never stored (lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16); 64 input channels."""
    _nhwc16(y_in, 'conv1x1_of_bn_relu_bn_stats')
    n, h, w_, cin = y_in.shape
    d = _lib.ConvDesc(n, h, w_, cin, cout, 1, 1, 0, 0, 0)
    ws = train_workspace(y_in.device)
    with torch.cuda.device(y_in.device):
        y = torch.empty((n, h, w_, cout), dtype=torch.float16, device=y_in.device)
        stats = torch.empty(2 * cout, dtype=torch.float32, device=y_in.device)
        check(lib().lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16(C.byref(d), ptr(y_in), ptr(in_stats), ptr(in_gamma), ptr(in_beta), ptr(y),
                                                             ptr(w_packed), ptr(bias), ptr(zero_line(y_in.device)), float(eps),
                                                             float(momentum), ptr(running_mean), ptr(running_var), ptr(ws), ws.numel(),
                                                             ptr(stats), stream_ptr()), 'lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16')
    return y, stats


def bn_train_apply(y, stats, gamma, beta, residual=None, relu=True):
    _nhwc16(y, 'bn_train_apply')
    c = y.size(3)
    z = torch.empty_like(y)
    with torch.cuda.device(y.device):
        check(lib().lfd_bn_train_apply_f16(ptr(y), y.numel() // c, c, ptr(stats), ptr(gamma), ptr(beta), ptr(residual),
                                           int(bool(relu)), ptr(z), stream_ptr()), 'lfd_bn_train_apply_f16')
    return z


def bn_train_backward(dz, y, z, stats, gamma, inv_scale, dgamma, dbeta, want_g=False, accumulate=False, relu=None,
                      beta=None):
    """-> (dy, g).  relu (default: z is not None): a ReLU follows the norm; its mask comes from the stored output z, or --
    z=None with relu=True and beta given, units without a residual input -- is recomputed from y.
    dgamma / dbeta: float32[C] outputs (unscaled; += if accumulate)."""
    relu = (z is not None) if relu is None else bool(relu)
    _nhwc16(dz, 'bn_train_backward')
    c = y.size(3)
    dy = torch.empty_like(y)
    g = torch.empty_like(y) if want_g else None
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_bn_train_bwd_f16(ptr(dz), ptr(y), ptr(z), int(relu), y.numel() // c, c, ptr(stats), ptr(gamma),
                                         ptr(beta), float(inv_scale), int(bool(accumulate)), ptr(ws), ws.numel(), ptr(dgamma),
                                         ptr(dbeta), ptr(dy), ptr(g),
                                         stream_ptr()), 'lfd_bn_train_bwd_f16')
    return dy, g


def gn_train_stats(y, groups, eps):
    _nhwc16(y, 'gn_train_stats')
    n, h, w_, c = y.shape
    stats = torch.empty(n * 2 * groups, dtype=torch.float32, device=y.device)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_gn_train_stats_f16(ptr(y), n, h * w_, c, groups, float(eps), ptr(ws), ws.numel(), ptr(stats),
                                           stream_ptr()), 'lfd_gn_train_stats_f16')
    return stats


def gn_train_apply(y, groups, stats, gamma, beta, relu=True):
    _nhwc16(y, 'gn_train_apply')
    n, h, w_, c = y.shape
    z = torch.empty_like(y)
    with torch.cuda.device(y.device):
        check(lib().lfd_gn_train_apply_f16(ptr(y), n, h * w_, c, groups, ptr(stats), ptr(gamma), ptr(beta), int(bool(relu)),
                                           ptr(z), stream_ptr()), 'lfd_gn_train_apply_f16')
    return z


def gn_train_stats_apply(y, groups, eps, gamma, beta, relu=True):
    """gn_train_stats + gn_train_apply as two launches (lfd_gn_train_stats_apply_f16) -> (stats, z)"""
    _nhwc16(y, 'gn_train_stats_apply')
    n, h, w_, c = y.shape
    stats = torch.empty(n * 2 * groups, dtype=torch.float32, device=y.device)
    z = torch.empty_like(y)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_gn_train_stats_apply_f16(ptr(y), n, h * w_, c, groups, float(eps), ptr(gamma), ptr(beta), int(bool(relu)),
                                                 ptr(ws), ws.numel(), ptr(stats), ptr(z), stream_ptr()), 'lfd_gn_train_stats_apply_f16')
    return stats, z


def _seg_array(seg_hw):
    return (C.c_int64 * len(seg_hw))(*[int(v) for v in seg_hw])


def gn_train_stats_apply_seg(y, seg_hw, groups, eps, gamma, beta, relu=True):
    """GroupNorm + ReLU of a LEVEL-CONCATENATED tensor y [n, P, c] (P = sum(seg_hw)), statistics per (image, segment)
    -> (stats [n * nseg, 2, groups], z) (lfd_gn_train_stats_apply_seg_f16)"""
    require_cuda(y, 'gn_train_stats_apply_seg')
    n, p, c = y.shape
    if y.dtype != torch.float16 or not y.is_contiguous() or p != sum(seg_hw):
        raise RuntimeError('gn_train_stats_apply_seg: contiguous fp16 [n, sum(seg_hw), c] expected')
    stats = torch.empty(n * len(seg_hw) * 2 * groups, dtype=torch.float32, device=y.device)
    z = torch.empty_like(y)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_gn_train_stats_apply_seg_f16(ptr(y), n, len(seg_hw), _seg_array(seg_hw), c, groups, float(eps), ptr(gamma),
                                                     ptr(beta), int(bool(relu)), ptr(ws), ws.numel(), ptr(stats), ptr(z),
                                                     stream_ptr()), 'lfd_gn_train_stats_apply_seg_f16')
    return stats, z


def gn_train_backward_seg(dz, y, z, seg_hw, groups, stats, gamma, inv_scale, dgamma, dbeta, accumulate=False):
    require_cuda(y, 'gn_train_backward_seg')
    n, p, c = y.shape
    dy = torch.empty_like(y)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_gn_train_bwd_seg_f16(ptr(dz), ptr(y), ptr(z), n, len(seg_hw), _seg_array(seg_hw), c, groups, ptr(stats),
                                             ptr(gamma), float(inv_scale), int(bool(accumulate)), ptr(ws), ws.numel(), ptr(dgamma),
                                             ptr(dbeta), ptr(dy), stream_ptr()), 'lfd_gn_train_bwd_seg_f16')
    return dy


def bn_train_apply_into(y, stats, gamma, beta, relu, z_concat, point0):
    """bn_train_apply of a level's tensor y [n, h, w, c] into rows [point0, point0 + h * w) of every image of z_concat [n, P, c]"""
    _nhwc16(y, 'bn_train_apply_into')
    n, h, w_, c = y.shape
    with torch.cuda.device(y.device):
        check(lib().lfd_bn_train_apply_into_f16(ptr(y), n, h * w_, c, ptr(stats), ptr(gamma), ptr(beta), int(bool(relu)),
                                                ptr(z_concat), z_concat.size(1), int(point0), stream_ptr()),
              'lfd_bn_train_apply_into_f16')


def bn_train_backward_from(dz_concat, point0, y, stats, gamma, beta, inv_scale, dgamma, dbeta, relu=True, accumulate=True):
    """bn_train_backward of a level's unit (no residual; ReLU mask recomputed from y) whose output gradient is rows
    [point0, point0 + h * w) of every image of dz_concat [n, P, c] -> dy [n, h, w, c]"""
    _nhwc16(y, 'bn_train_backward_from')
    n, h, w_, c = y.shape
    dy = torch.empty_like(y)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_bn_train_bwd_from_f16(ptr(dz_concat), dz_concat.size(1), int(point0), ptr(y), int(bool(relu)), n, h * w_, c,
                                              ptr(stats), ptr(gamma), ptr(beta), float(inv_scale), int(bool(accumulate)), ptr(ws),
                                              ws.numel(), ptr(dgamma), ptr(dbeta), ptr(dy), stream_ptr()),
              'lfd_bn_train_bwd_from_f16')
    return dy


def conv2d_bn_partials(x, w_packed, bias, cin, cout, ks, stride, rows):
    """conv2d_bn_stats without its final pass: -> (y, nrows), the per-workgroup statistics rows in `rows` (fp32, >= 512 * 2 * cout);
    None when the shape has no statistics kernel (lfd_conv2d_bn_partials_nhwc_f16)"""
    _nhwc16(x, 'conv2d_bn_partials')
    n, h, w_, c = x.shape
    pad = ks // 2
    oh, ow = (h + 2 * pad - ks) // stride + 1, (w_ + 2 * pad - ks) // stride + 1
    d = _lib.ConvDesc(n, h, w_, cin, cout, ks, stride, 0, 0, 0)
    nrows = C.c_int32(0)
    with torch.cuda.device(x.device):
        y = torch.empty((n, oh, ow, cout), dtype=torch.float16, device=x.device)
        rc = lib().lfd_conv2d_bn_partials_nhwc_f16(C.byref(d), ptr(x), ptr(y), ptr(w_packed), ptr(bias), ptr(zero_line(x.device)),
                                                   ptr(rows), rows.numel() * 4, C.byref(nrows), stream_ptr())
    if rc == -4:          # LFD_ERR_UNSUPPORTED: no STATS kernel for this shape
        return None
    check(rc, 'lfd_conv2d_bn_partials_nhwc_f16')
    return y, int(nrows.value)


def bn_train_finish_into_levels(levels, n, relu, z_concat):
    """the per-channel finals and the apply passes of several units whose convs left statistics rows (conv2d_bn_partials), two
    launches for all of them; levels: [(point0, y, rows, nrows, eps, momentum, running_mean, running_var, gamma, beta)] -> [stats]
    (lfd_bn_train_finish_into_levels_f16)"""
    arr = (_lib.BnFwdLevel * len(levels))()
    out = []
    for l, (point0, y, rows, nrows, eps, momentum, rm, rv, gamma, beta) in enumerate(levels):
        st = torch.empty(2 * y.size(3), dtype=torch.float32, device=y.device)
        out.append(st)
        a = arr[l]
        a.y, a.rows, a.running_mean, a.running_var, a.stats, a.gamma, a.beta = ptr(y), ptr(rows), ptr(rm), ptr(rv), ptr(st), ptr(gamma), ptr(beta)
        a.hw, a.point0, a.channels, a.nrows, a.eps, a.momentum = y.size(1) * y.size(2), int(point0), y.size(3), int(nrows), float(eps), float(momentum)
    with torch.cuda.device(z_concat.device):
        check(lib().lfd_bn_train_finish_into_levels_f16(arr, len(levels), int(n), int(bool(relu)), ptr(z_concat), z_concat.size(1),
                                                        stream_ptr()), 'lfd_bn_train_finish_into_levels_f16')
    return out


def bn_train_backward_from_levels(dz_concat, levels, inv_scale, relu=True, accumulate=True):
    """bn_train_backward_from for several levels in three launches; levels: [(point0, y, stats, gamma, beta, dgamma, dbeta)]
    -> [dy per level] (lfd_bn_train_bwd_from_levels_f16; bit-identical to the per-level calls)"""
    require_cuda(dz_concat, 'bn_train_backward_from_levels')
    arr = (_lib.BnBwdLevel * len(levels))()
    dys = []
    n = dz_concat.size(0)
    for l, (point0, y, stats, gamma, beta, dgamma, dbeta) in enumerate(levels):
        _nhwc16(y, 'bn_train_backward_from_levels')
        dy = torch.empty_like(y)
        dys.append(dy)
        a = arr[l]
        a.y, a.dy, a.stats, a.gamma, a.beta, a.dgamma, a.dbeta = ptr(y), ptr(dy), ptr(stats), ptr(gamma), ptr(beta), ptr(dgamma), ptr(dbeta)
        a.hw, a.point0, a.channels = y.size(1) * y.size(2), int(point0), y.size(3)
    ws = train_workspace(dz_concat.device)
    with torch.cuda.device(dz_concat.device):
        check(lib().lfd_bn_train_bwd_from_levels_f16(ptr(dz_concat), dz_concat.size(1), arr, len(levels), int(bool(relu)), n,
                                                     float(inv_scale), int(bool(accumulate)), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_bn_train_bwd_from_levels_f16')
    return dys


def gn_train_backward(dz, y, z, groups, stats, gamma, inv_scale, dgamma, dbeta, accumulate=False):
    _nhwc16(dz, 'gn_train_backward')
    n, h, w_, c = y.shape
    dy = torch.empty_like(y)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        check(lib().lfd_gn_train_bwd_f16(ptr(dz), ptr(y), ptr(z), n, h * w_, c, groups, ptr(stats), ptr(gamma),
                                         float(inv_scale), int(bool(accumulate)), ptr(ws), ws.numel(), ptr(dgamma),
                                         ptr(dbeta), ptr(dy), stream_ptr()), 'lfd_gn_train_bwd_f16')
    return dy


def conv3x3s2_dgrad(dy, w_packed_dgrad, h, w, residual=None):
    """data gradient of a 3x3 stride-2 pad-1 conv, 64 -> 64 channels: dx [N,h,w,64] from dy [N,(h-1)//2+1,(w-1)//2+1,64] without
    the zero-inserted tensor (csrc/dgrad_s2.hip); bit-identical to zero_insert2 + conv2d_nhwc on the same packed filter"""
    _nhwc16(dy, 'conv3x3s2_dgrad')
    n, ho, wo, c = dy.shape
    if c != 64 or ho != (h - 1) // 2 + 1 or wo != (w - 1) // 2 + 1:
        raise RuntimeError('conv3x3s2_dgrad: dy must be [N, (h-1)//2+1, (w-1)//2+1, 64]')
    with torch.cuda.device(dy.device):
        dx = torch.empty((n, h, w, 64), dtype=torch.float16, device=dy.device)
        check(lib().lfd_conv3x3s2_dgrad_nhwc_f16(n, h, w, ptr(dy), ptr(dx), ptr(w_packed_dgrad), ptr(residual), stream_ptr()),
              'lfd_conv3x3s2_dgrad_nhwc_f16')
    return dx


def zero_insert2(t, ho, wo):
    _nhwc16(t, 'zero_insert2')
    n, hi, wi, c = t.shape
    out = torch.empty((n, ho, wo, c), dtype=torch.float16, device=t.device)
    with torch.cuda.device(t.device):
        check(lib().lfd_zero_insert2_nhwc_f16(ptr(t), n, hi, wi, c, ho, wo, ptr(out), stream_ptr()),
              'lfd_zero_insert2_nhwc_f16')
    return out


def pack_conv_weight_train(weight, data_gradient=False, rows=None):
    """nn.Conv2d.weight (fp32 OIHW) -> packed fp16 fragments in one launch (csrc/train.hip k_pack_weight): the forward
    conv (optionally zero-padded to `rows` output rows) or the data-gradient conv (roles swapped, taps flipped)."""
    require_cuda(weight, 'pack_conv_weight_train')
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.contiguous().float()
    cout, cin, ks, _ = w.shape
    rows = cout if rows is None else int(rows)
    lc, li = (cin, cout) if data_gradient else (rows, cin)
    out = torch.empty((lc // 32, ks * ks * (li // 16), 64, 8), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        if data_gradient:
            check(lib().lfd_pack_conv_weight_train_f16(ptr(w), cout, cin, ks, 1, lc, ptr(out), stream_ptr()),
                  'lfd_pack_conv_weight_train_f16')
        else:
            check(lib().lfd_pack_conv_weight_train_f16(ptr(w), rows, cin, ks, 0, cout, ptr(out), stream_ptr()),
                  'lfd_pack_conv_weight_train_f16')
    return out


class PackBatch(object):
    """All weight packs of a pass in ONE launch (lfd_pack_conv_weights_train_f16): the job table (weight pointer, output
    pointer, shape) is built and uploaded once -- the parameters live in a flat buffer, their addresses do not change --
    and `run()` re-packs the current values.  outs[i] is the packed fp16 tensor of weights[i]."""

    def __init__(self, weights, data_gradient):
        dev = weights[0].device
        require_cuda(weights[0], 'PackBatch')
        self.key = tuple(w.data_ptr() for w in weights)
        self.outs = []
        jobs = (_lib.PackJob * len(weights))()
        first = 0
        for i, w in enumerate(weights):
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise RuntimeError('PackBatch: contiguous float32 OIHW weights expected')
            cout, cin, ks, _ = w.shape
            lc, li = (cin, cout) if data_gradient else (cout, cin)
            out = torch.empty((lc // 32, ks * ks * (li // 16), 64, 8), dtype=torch.float16, device=dev)
            j = jobs[i]
            j.w, j.out = w.data_ptr(), out.data_ptr()
            j.cout, j.cin, j.ks, j.mode = cout, cin, ks, int(bool(data_gradient))
            j.rows_valid, j.first_vec = (lc if data_gradient else cout), first
            first += out.numel() // 8
            self.outs.append(out)
        self.total, self.njobs = first, len(weights)
        self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.device = dev

    def matches(self, weights):
        return self.key == tuple(w.data_ptr() for w in weights)

    def run(self):
        with torch.cuda.device(self.device):
            check(lib().lfd_pack_conv_weights_train_f16(ptr(self.table), self.njobs, self.total, stream_ptr()),
                  'lfd_pack_conv_weights_train_f16')
        return self.outs


def conv_wgrad(x, dy, ks, stride, inv_scale, out=None, accumulate=False):
    """dW [cout,cin,ks,ks] fp32 of conv(x, W) with pad ks//2 given dL/dy (scaled by 1/inv_scale); += into `out` if
    accumulate."""
    _nhwc16(x, 'conv_wgrad')
    _nhwc16(dy, 'conv_wgrad')
    n, h, w_, cin = x.shape
    cout = dy.size(3)
    if out is None:
        out = torch.empty((cout, cin, ks, ks), dtype=torch.float32, device=x.device)
    ws = train_workspace(x.device)
    with torch.cuda.device(x.device):
        check(lib().lfd_conv_wgrad_nhwc_f16(ptr(x), ptr(dy), n, h, w_, cin, cout, ks, stride, float(inv_scale),
                                            int(bool(accumulate)), ptr(ws), ws.numel(), ptr(out), stream_ptr()),
              'lfd_conv_wgrad_nhwc_f16')
    return out


def conv_wgrad_partial_floats(x, dy, ks, stride):
    """floats of the partial buffer lfd_conv_wgrad_partials_nhwc_f16 fills for these shapes -> (floats, workgroup rows, blocks)"""
    n, h, w_, cin = x.shape
    cout = dy.size(3)
    nwg = lib().lfd_conv_wgrad_partial_rows(n, h, w_, cin, cout, ks, stride)
    if nwg < 0:
        check(nwg, 'lfd_conv_wgrad_partial_rows')
    nblk = ((cout + 63) // 64) * ((cin + 63) // 64)
    return nwg * nblk * ks * ks * 4096, nwg, nblk


def conv_wgrad_partials(x, dy, ks, stride, partials):
    """first stage of conv_wgrad only: the per-workgroup partial sums of dW into `partials` (fp32, conv_wgrad_partial_floats);
    the sums are taken later, for all convs of the iteration at once, by WgradFinals.launch"""
    _nhwc16(x, 'conv_wgrad_partials')
    _nhwc16(dy, 'conv_wgrad_partials')
    n, h, w_, cin = x.shape
    with torch.cuda.device(x.device):
        check(lib().lfd_conv_wgrad_partials_nhwc_f16(ptr(x), ptr(dy), n, h, w_, cin, dy.size(3), ks, stride, ptr(partials),
                                                     partials.numel() * 4, stream_ptr()), 'lfd_conv_wgrad_partials_nhwc_f16')


def conv1x1_wgrad_partials_of_bn_relu(y_in, in_stats, in_gamma, in_beta, dy, partials):
    """conv_wgrad_partials(ks 1, stride 1) with x = relu(BatchNorm(y_in)) re-formed in the tile loader (the operand of a
    conv1x1_of_bn_relu_bn_stats forward was never stored)"""
    _nhwc16(y_in, 'conv1x1_wgrad_partials_of_bn_relu')
    _nhwc16(dy, 'conv1x1_wgrad_partials_of_bn_relu')
    n, h, w_, cin = y_in.shape
    with torch.cuda.device(y_in.device):
        check(lib().lfd_conv1x1_wgrad_partials_of_bn_relu_f16(ptr(y_in), ptr(in_stats), ptr(in_gamma), ptr(in_beta), ptr(dy), n, h, w_,
                                                              cin, dy.size(3), ptr(partials), partials.numel() * 4, stream_ptr()),
              'lfd_conv1x1_wgrad_partials_of_bn_relu_f16')


class WgradFinals(object):
    """The deferred final stage of an iteration's weight gradients and of the small per-level gradient copies: collects jobs
    during the backward pass, launches lfd_wgrad_final_batched_f32 + lfd_rows_sum_batched_f32 once.  The device tables are
    cached by content (pointers and shapes repeat from iteration to iteration: flat gradient buffers, persistent partial
    buffers), so a steady-state backward uploads nothing -- which is also what makes it capturable into a HIP graph."""

    def __init__(self, device):
        self.device = device
        self.cache = {}
        self.reset()

    def reset(self):
        self.wj, self.rj, self.chains = [], [], {}

    def add_wgrad(self, partials, nwg, nblk, cin, cout, taps, inv_scale, targets):
        """targets: [(dw tensor, co_lo, co_hi)] -- the rows of this conv that go to each parameter gradient (+=)"""
        for dw, lo, hi in targets:
            key = (dw.data_ptr(), lo, hi)
            idx = len(self.wj)
            self.wj.append(dict(partials=partials.data_ptr(), dw=dw.data_ptr(), nwg=nwg, nblk=nblk, cin=cin, cout=cout, taps=taps,
                                co_lo=lo, co_hi=hi, head=key not in self.chains, next=-1, inv_scale=float(inv_scale)))
            if key in self.chains:
                self.wj[self.chains[key]]['next'] = idx
            self.chains[key] = idx

    def add_rowsum(self, src, nrows, row_stride, count, dst, accumulate=True):
        self.rj.append((src.data_ptr(), dst.data_ptr(), int(nrows), int(row_stride), int(count), int(bool(accumulate))))

    def _table(self, kind, key, build):
        """One device table per CONTENT, kept for the life of the plan: a captured training graph holds the raw address of
        the table that was live at capture, and the content key contains the loss scale (inv_scale) -- with a DynamicLossScale
        an overflow halves the scale, an eager iteration builds a second table, and `growth_interval` iterations later the
        first graph replays again.  Replacing (= freeing) the old table handed that replay a dangling job table.
        A table is ~64 B per job; the keys of a run are a handful (loss scales x batch shapes)."""
        ent = self.cache.get((kind, key))
        if ent is None:
            raw = build()
            ent = (torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).to(self.device), raw)
            self.cache[(kind, key)] = ent
        return ent[0]

    def launch(self):
        with torch.cuda.device(self.device):
            if self.wj:
                first = 0
                for j in self.wj:
                    j['first_block'] = first if j['head'] else -1
                    if j['head']:
                        first += j['nblk'] * j['taps'] * 32
                key = tuple(tuple(sorted(j.items())) for j in self.wj)

                def build():
                    arr = (_lib.WgradJob * len(self.wj))()
                    for a, j in zip(arr, self.wj):
                        a.partials, a.dw, a.nwg, a.nblk, a.cin, a.cout, a.taps = (j['partials'], j['dw'], j['nwg'], j['nblk'], j['cin'],
                                                                                  j['cout'], j['taps'])
                        a.co_lo, a.co_hi, a.first_block, a.next, a.accumulate, a.inv_scale = (j['co_lo'], j['co_hi'], j['first_block'],
                                                                                              j['next'], 1, j['inv_scale'])
                    return arr
                tab = self._table('w', key, build)
                check(lib().lfd_wgrad_final_batched_f32(ptr(tab), len(self.wj), first, stream_ptr()), 'lfd_wgrad_final_batched_f32')
            if self.rj:
                key = tuple(self.rj)

                def build_r():
                    arr = (_lib.RowsumJob * len(self.rj))()
                    for a, j in zip(arr, self.rj):
                        a.src, a.dst, a.nrows, a.row_stride, a.count, a.accumulate = j
                    return arr
                tab = self._table('r', key, build_r)
                check(lib().lfd_rows_sum_batched_f32(ptr(tab), len(self.rj), stream_ptr()), 'lfd_rows_sum_batched_f32')
        self.reset()


def _head_out_segs(segs, field, tensors):
    arr = (_lib.HeadOutSeg * len(segs))()
    for i, (sg, t) in enumerate(zip(segs, tensors)):
        arr[i].channels, arr[i].row0 = int(sg['channels']), int(sg['row0'])
        arr[i].scale = ptr(sg.get('scale'))
        setattr(arr[i], field, ptr(t))
        if field == 'grad':
            arr[i].dbias, arr[i].dscale = ptr(sg.get('dbias')), ptr(sg.get('dscale'))
    return arr


def head_out_split_concat(y_concat, hw, segs, outs, point0):
    """head_out_split for a level whose conv output is rows [point0, point0 + hw) of every image of y_concat [n, P, 64]"""
    require_cuda(y_concat, 'head_out_split_concat')
    n = y_concat.size(0)
    arr = _head_out_segs(segs, 'out', outs)
    with torch.cuda.device(y_concat.device):
        check(lib().lfd_head_out_split_concat_f16(ptr(y_concat), n, int(hw), outs[0].size(1), int(point0), arr, len(segs),
                                                  stream_ptr()), 'lfd_head_out_split_concat_f16')


def head_out_grad_concat(y_concat, hw, segs, grads, point0, loss_scale, dy_concat):
    """head_out_grad writing the level's rows of dy_concat [n, P, 64]; accumulates dbias / dscale of the segments"""
    require_cuda(y_concat, 'head_out_grad_concat')
    n = y_concat.size(0)
    arr = _head_out_segs(segs, 'grad', grads)
    ws = train_workspace(y_concat.device)
    with torch.cuda.device(y_concat.device):
        check(lib().lfd_head_out_grad_concat_f16(ptr(y_concat), n, int(hw), grads[0].size(1), int(point0), arr, len(segs),
                                                 float(loss_scale), ptr(dy_concat), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_head_out_grad_concat_f16')


def _head_out_levels(levels, field):
    """levels: [(hw, point0, segs, tensors)] -> (lfd_head_out_level_t array, keep-alive list)"""
    arr = (_lib.HeadOutLevel * len(levels))()
    for l, (hw, point0, segs, tensors) in enumerate(levels):
        arr[l].hw, arr[l].nsegs, arr[l].point0 = int(hw), len(segs), int(point0)
        sa = _head_out_segs(segs, field, tensors)
        for i in range(len(segs)):
            arr[l].segs[i] = sa[i]
    return arr


def head_out_split_levels(y_concat, levels):
    """head_out_split_concat for ALL pyramid levels of one output conv in one launch; levels: [(hw, point0, segs, outs)]
    (lfd_head_out_split_levels_f16; bit-identical to the per-level calls)"""
    require_cuda(y_concat, 'head_out_split_levels')
    arr = _head_out_levels(levels, 'out')
    with torch.cuda.device(y_concat.device):
        check(lib().lfd_head_out_split_levels_f16(ptr(y_concat), y_concat.size(0), levels[0][3][0].size(1), arr, len(levels),
                                                  stream_ptr()), 'lfd_head_out_split_levels_f16')


def head_out_grad_levels(y_concat, levels, loss_scale, dy_concat):
    """head_out_grad_concat for ALL pyramid levels in one launch (+ one final launch): levels: [(hw, point0, segs, grads)], segs
    with their dbias / dscale targets (lfd_head_out_grad_levels_f16; bit-identical to the per-level calls)"""
    require_cuda(y_concat, 'head_out_grad_levels')
    arr = _head_out_levels(levels, 'grad')
    ws = train_workspace(y_concat.device)
    with torch.cuda.device(y_concat.device):
        check(lib().lfd_head_out_grad_levels_f16(ptr(y_concat), y_concat.size(0), levels[0][3][0].size(1), arr, len(levels),
                                                 float(loss_scale), ptr(dy_concat), ptr(ws), ws.numel(), stream_ptr()),
              'lfd_head_out_grad_levels_f16')


def head_out_split(y, segs, outs, point0):
    """y [n,h,w,64] fp16 (a level's padded output conv) -> outs[i][:, point0:point0+h*w, :] = float(y[..., rows of segment i])
    (* scale); outs[i]: the level-concatenated [n, P, channels] fp32 tensors.  segs: dicts channels, row0, scale (tensor or
    None).  (lfd_head_out_split_f16)"""
    _nhwc16(y, 'head_out_split')
    n, h, w_, rows = y.shape
    if rows != 64:
        raise RuntimeError('head_out_split: 64 output rows expected')
    for sg, o in zip(segs, outs):
        if o.dtype != torch.float32 or not o.is_contiguous() or o.size(0) != n or o.size(2) != sg['channels']:
            raise RuntimeError('head_out_split: outs must be contiguous fp32 [n, P, channels]')
    arr = _head_out_segs(segs, 'out', outs)
    with torch.cuda.device(y.device):
        check(lib().lfd_head_out_split_f16(ptr(y), n, h * w_, outs[0].size(1), int(point0), arr, len(segs), stream_ptr()),
              'lfd_head_out_split_f16')


def head_out_grad(y, segs, grads, point0, loss_scale):
    """-> dy [n,h,w,64] fp16 = grads[i][:, point0:point0+h*w, :] (* scale) * loss_scale in the rows of segment i, zero
    elsewhere; accumulates segs[i]['dbias'] / ['dscale'] (fp32 tensors or None) in place.  (lfd_head_out_grad_f16)"""
    _nhwc16(y, 'head_out_grad')
    n, h, w_, rows = y.shape
    if rows != 64:
        raise RuntimeError('head_out_grad: 64 output rows expected')
    for sg, g in zip(segs, grads):
        if g.dtype != torch.float32 or not g.is_contiguous() or g.size(0) != n or g.size(2) != sg['channels']:
            raise RuntimeError('head_out_grad: grads must be contiguous fp32 [n, P, channels]')
        for k in ('dbias', 'dscale'):
            t = sg.get(k)
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError('head_out_grad: %s must be contiguous fp32' % k)
    arr = _head_out_segs(segs, 'grad', grads)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        dy = torch.empty_like(y)
        check(lib().lfd_head_out_grad_f16(ptr(y), n, h * w_, grads[0].size(1), int(point0), arr, len(segs), float(loss_scale),
                                          ptr(dy), ptr(ws), ws.numel(), stream_ptr()), 'lfd_head_out_grad_f16')
    return dy


def stem_conv0_train_fwd(x_nchw, weight):
    require_cuda(x_nchw, 'stem_conv0_train_fwd')
    x = x_nchw.contiguous().float()
    n, _, h, w_ = x.shape
    c = weight.size(0)
    y = torch.empty((n, (h + 1) // 2, (w_ + 1) // 2, c), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().lfd_stem_conv0_train_fwd(ptr(x), n, h, w_, c, ptr(weight.detach().contiguous().float()), ptr(y),
                                             stream_ptr()), 'lfd_stem_conv0_train_fwd')
    return y


def stem_conv0_train_fwd_bn_stats(x_nchw, weight, eps, momentum, running_mean=None, running_var=None):
    """stem_conv0_train_fwd + the batch statistics of its output from the conv's own stores -> (y, float32[2*C])"""
    require_cuda(x_nchw, 'stem_conv0_train_fwd_bn_stats')
    x = x_nchw.contiguous().float()
    n, _, h, w_ = x.shape
    c = weight.size(0)
    ws = train_workspace(x.device)
    with torch.cuda.device(x.device):
        y = torch.empty((n, (h + 1) // 2, (w_ + 1) // 2, c), dtype=torch.float16, device=x.device)
        stats = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        check(lib().lfd_stem_conv0_train_fwd_bn_stats(ptr(x), n, h, w_, c, ptr(weight.detach().contiguous().float()), ptr(y),
                                                      float(eps), float(momentum), ptr(running_mean), ptr(running_var), ptr(ws),
                                                      ws.numel(), ptr(stats), stream_ptr()), 'lfd_stem_conv0_train_fwd_bn_stats')
    return y, stats


def stem_conv0_bn_bwd_wgrad(x_nchw, dz, y, stats, gamma, beta, inv_scale, dgamma, dbeta, dw, sum_rows=0):
    """backward of the first conv unit (conv 3 -> 64 + train-mode BatchNorm + ReLU) from dz = dL/d(output): dgamma / dbeta / dw += ;
    no dy tensor (lfd_stem_conv0_bn_bwd_wgrad).  sum_rows > 0: BatchNorm's partial sums are already in the training workspace
    (conv1x1_dgrad_bn_bwd_sums left them): no pass over (dz, y) for them."""
    require_cuda(x_nchw, 'stem_conv0_bn_bwd_wgrad')
    x = x_nchw.contiguous().float()
    n, _, h, w_ = x.shape
    c = y.size(3)
    ws = train_workspace(x.device)
    with torch.cuda.device(x.device):
        check(lib().lfd_stem_conv0_bn_bwd_wgrad_rows(ptr(x), ptr(dz), ptr(y), n, h, w_, c, ptr(stats), ptr(gamma), ptr(beta),
                                                     float(inv_scale), 1, int(sum_rows), ptr(ws), ws.numel(), ptr(dgamma), ptr(dbeta),
                                                     ptr(dw), stream_ptr()), 'lfd_stem_conv0_bn_bwd_wgrad_rows')


def conv1x1_dgrad_bn_bwd_sums(dy, w_packed_dgrad, zero_bias, y_unit, unit_stats, unit_gamma, unit_beta):
    """The data gradient of a 1x1 stride-1 conv 64 -> 64 whose input was the activation of a BatchNorm + ReLU unit (no residual),
    with that unit's backward sums taken in the conv's epilogue -> (dz, sum_rows): the rows stay in the training workspace for
    the NEXT call, which must be bn_train_backward_rows / stem_conv0_bn_bwd_wgrad(sum_rows=...) of that unit
    (lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16)."""
    _nhwc16(dy, 'conv1x1_dgrad_bn_bwd_sums')
    _nhwc16(y_unit, 'conv1x1_dgrad_bn_bwd_sums')
    n, h, w_, c = dy.shape
    if c != 64 or y_unit.shape != dy.shape:
        raise RuntimeError('conv1x1_dgrad_bn_bwd_sums: 64 -> 64 channels, dy and y of one shape')
    d = _lib.ConvDesc(n, h, w_, 64, 64, 1, 1, 0, 0, 0)
    ws = train_workspace(dy.device)
    rows = C.c_int32(0)
    with torch.cuda.device(dy.device):
        dz = torch.empty_like(dy)
        check(lib().lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16(C.byref(d), ptr(dy), ptr(dz), ptr(w_packed_dgrad), ptr(zero_bias),
                                                           ptr(zero_line(dy.device)), ptr(y_unit), ptr(unit_stats), ptr(unit_gamma),
                                                           ptr(unit_beta), ptr(ws), ws.numel(), C.byref(rows), stream_ptr()),
              'lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16')
    return dz, int(rows.value)


def bn_train_backward_rows(dz, y, stats, gamma, beta, inv_scale, dgamma, dbeta, sum_rows, accumulate=True):
    """bn_train_backward (ReLU mask recomputed from y, no g) whose partial sums conv1x1_dgrad_bn_bwd_sums already left in the
    training workspace -> dy (lfd_bn_train_bwd_rows_f16)"""
    _nhwc16(dz, 'bn_train_backward_rows')
    c = y.size(3)
    ws = train_workspace(y.device)
    with torch.cuda.device(y.device):
        dy = torch.empty_like(y)
        check(lib().lfd_bn_train_bwd_rows_f16(ptr(dz), ptr(y), y.numel() // c, c, ptr(stats), ptr(gamma), ptr(beta), float(inv_scale),
                                              int(bool(accumulate)), int(sum_rows), ptr(ws), ws.numel(), ptr(dgamma), ptr(dbeta),
                                              ptr(dy), stream_ptr()), 'lfd_bn_train_bwd_rows_f16')
    return dy


def stem_conv0_wgrad(x_nchw, dy, inv_scale, out=None, accumulate=False):
    x = x_nchw.contiguous().float()
    n, _, h, w_ = x.shape
    c = dy.size(3)
    if out is None:
        out = torch.empty((c, 3, 3, 3), dtype=torch.float32, device=x.device)
    ws = train_workspace(x.device)
    with torch.cuda.device(x.device):
        check(lib().lfd_stem_conv0_wgrad(ptr(x), ptr(dy), n, h, w_, c, float(inv_scale), int(bool(accumulate)), ptr(ws),
                                         ws.numel(), ptr(out), stream_ptr()), 'lfd_stem_conv0_wgrad')
    return out
