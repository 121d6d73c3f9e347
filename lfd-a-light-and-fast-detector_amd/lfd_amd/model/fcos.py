"""FCOS meta-architecture -- host-side mirror of lfd/model/fcos.py:12-449 (class FCOS) and :452-900 (FCOSv1, the
multi-label experiment: same network and post-processing, per-class binary targets).

Same constructor kwargs, `forward(x) -> (cls [N,P,C], reg [N,P,4] distances, centerness [N,P,1])` fp32,
`head_indexes_to_feature_map_sizes`, `get_param_groups_for_optimizer`, `annotation_to_target`, `centerness_target`,
`distance2bbox`, `get_loss`, `get_results` contracts.  Where the arithmetic runs:

  eval-mode forward   engine_sibling.sibling_forward: backbone on the fused LFD kernels, neck / head layer by layer on
                      the MFMA conv, GroupNorm, upsample-add and output-packing kernels (csrc/conv_impl.h, sibling.hip)
  get_results         ONE device pass for the whole batch, lfd_detect_batched_ex (csrc/postproc.hip): sigmoid scores x
                      sigmoid centerness, per-level pre-NMS top-k, decode, threshold, class-wise NMS, post-NMS cap --
                      instead of the reference's per-image, per-level Python loop (fcos.py:331-412)
  get_loss            targets with the reference's [P,G] tensor algebra on the host (where the reference computes them
                      too, :108-209), losses on the HIP loss kernels (focal / IoU-family / BCE, csrc/losses.hip, boxloss.hip)
  train-mode forward  PyTorch-ROCm autograd over the same parameters (training-only route, as for LFD)
"""
import torch
import torch.nn as nn

from .. import _lib, engine_sibling, ops, train_engine
from .lfd import LFD
from .utils import multiclass_nms  # noqa: F401  (importable like the reference module)

__all__ = ['FCOS', 'FCOSv1']

INF = 1e8


class FCOS(nn.Module):

    def __init__(self, backbone=None, neck=None, head=None, num_classes=80,
                 regress_ranges=((0, 64), (64, 128), (128, 256), (256, 512), (512, INF)),
                 point_strides=(8, 16, 32, 64, 128), classification_loss_func=None, regression_loss_func=None,
                 centerness_loss_func=None, classification_threshold=0.05, nms_threshold=0.5, pre_nms_bbox_limit=1000,
                 post_nms_bbox_limit=100, param_groups_cfg=None):
        super().__init__()
        assert len(regress_ranges) == len(point_strides), 'the length should be the same!'
        self._backbone, self._neck, self._head = backbone, neck, head
        self._num_classes = num_classes           # foreground labels 0..C-1, background = C
        self._regress_ranges = regress_ranges
        self._point_strides = point_strides
        self._num_levels = len(point_strides)
        self._classification_loss_func = classification_loss_func
        self._regression_loss_func = regression_loss_func
        self._centerness_loss_func = centerness_loss_func
        self._classification_threshold = classification_threshold
        self._nms_cfg = dict(type='nms', iou_thr=nms_threshold)
        self._pre_nms_bbox_limit = pre_nms_bbox_limit
        self._post_nms_bbox_limit = post_nms_bbox_limit
        self._param_groups_cfg = param_groups_cfg
        self._head_indexes_to_feature_map_sizes = dict()
        self.max_candidates = 8192
        self.use_graph = False       # capture the eval-mode forward into a HIP graph per input buffer

    @property
    def head_indexes_to_feature_map_sizes(self):
        return self._head_indexes_to_feature_map_sizes

    def get_param_groups_for_optimizer(self):
        """fcos.py:53-80: biases of everything that is not a norm layer form their own group (bias_lr / bias_weight_decay)."""
        cfg = self._param_groups_cfg
        if cfg is None:
            return self.parameters()
        assert isinstance(cfg, dict)
        biases, others = [], []
        for _, module in self.named_modules():
            is_norm = isinstance(module, (nn.BatchNorm2d, nn.GroupNorm))
            for pname, p in module.named_parameters(recurse=False):
                (biases if (not is_norm and 'bias' in pname) else others).append(p)
        group = dict(params=biases)
        for src, dst in (('bias_lr', 'lr'), ('bias_weight_decay', 'weight_decay')):
            if src in cfg:
                group[dst] = cfg[src]
        return [dict(params=others), group]

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        """fcos.py:414-449"""
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_train(x)
        cls, reg, ctr, sizes = engine_sibling.sibling_forward(self, x, use_graph=self.use_graph)
        for i, hw in enumerate(sizes):
            self._head_indexes_to_feature_map_sizes[i] = hw
        return cls, reg, ctr

    def _forward_train(self, x):
        _lib.require_cuda(x, 'FCOS.forward')
        if train_engine.supported(self._backbone):
            feats = list(train_engine.backbone_train_forward(self._backbone, x))
        else:
            feats = LFD._backbone_train_torch(self, x)
        cls_l, reg_l, ctr_l = self._head(self._neck(feats))
        outs = []
        for i, c in enumerate(cls_l):
            self._head_indexes_to_feature_map_sizes[i] = (c.shape[2], c.shape[3])
        for maps in (cls_l, reg_l, ctr_l):
            outs.append(torch.cat([m.permute(0, 2, 3, 1).reshape(m.shape[0], -1, m.shape[1]) for m in maps], dim=1))
        return tuple(outs)

    # ------------------------------------------------------------------ points / targets
    def generate_point_coordinates(self, feature_map_sizes):
        """fcos.py:82-106: x = j*stride, y = i*stride (the reference dropped mmdet's half-stride offset), int64 CPU"""
        assert len(feature_map_sizes) == len(self._point_strides)
        out = []
        for i, s in enumerate(self._point_strides):
            h, w = feature_map_sizes[i]
            ys, xs = torch.meshgrid(torch.arange(0, h * s, s), torch.arange(0, w * s, s), indexing='ij')
            out.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), dim=-1))
        return out

    def annotation_to_target(self, all_point_coordinates_list, gt_bboxes_list, gt_labels_list):
        """fcos.py:108-209 -> (labels [N,P] with C = background, distances [N,P,4]); a point belongs to the smallest-area
        box that contains it strictly and whose largest distance lies in the level's regress range."""
        ranges = torch.cat([pts.new_tensor(self._regress_ranges[i])[None].expand_as(pts)
                            for i, pts in enumerate(all_point_coordinates_list)], dim=0)
        points = torch.cat(all_point_coordinates_list, dim=0)
        labels, dists = [], []
        for boxes, lab in zip(gt_bboxes_list, gt_labels_list):
            l, d = self._targets_single(boxes, lab, points, ranges)
            labels.append(l)
            dists.append(d)
        return torch.stack(labels, dim=0), torch.stack(dists, dim=0)

    def _targets_single(self, gt, gt_labels, points, ranges):
        assert gt.size(0) == gt_labels.size(0)
        P, G = points.size(0), gt_labels.size(0)
        if G == 0:
            return self._empty_labels(gt_labels, P), gt.new_zeros((P, 4))
        areas = (gt[:, 2] * gt[:, 3])[None].repeat(P, 1)
        rng = ranges[:, None, :].expand(P, G, 2)
        gb = gt[None].expand(P, G, 4)
        px = points[:, 0][:, None].expand(P, G)
        py = points[:, 1][:, None].expand(P, G)
        dist = torch.stack((px - gb[:, :, 0], py - gb[:, :, 1],
                            (gb[:, :, 0] + gb[:, :, 2] - 1) - px, (gb[:, :, 1] + gb[:, :, 3] - 1) - py), dim=-1)
        inside = dist.min(dim=-1)[0] > 0
        far = dist.max(dim=-1)[0]
        valid = inside & (far >= rng[:, :, 0]) & (far <= rng[:, :, 1])
        areas = areas * valid + INF * (~valid)
        best_area, best = areas.min(dim=1)
        return self._labels(gt_labels, valid, best, best_area), dist[range(P), best]

    def _empty_labels(self, gt_labels, P):
        return gt_labels.new_full((P,), self._num_classes)

    def _labels(self, gt_labels, valid, best, best_area):
        """fcos.py:178-181: the label of the smallest valid box, C (background) where no box is valid"""
        return gt_labels[best] * (best_area != INF) + self._num_classes * (best_area == INF)

    def _flatten_classification(self, pred_cls, cls_t, dev):
        """-> (logits [M, C'], integer targets [M], indices of the positive POINTS)   (fcos.py:264-284)"""
        ct = cls_t.reshape(-1).to(dev)
        return pred_cls.reshape(-1, self._num_classes), ct, (ct != self._num_classes).nonzero().reshape(-1)

    def centerness_target(self, pos_flatten_regress_targets):
        """fcos.py:211-215"""
        lr = pos_flatten_regress_targets[:, [0, 2]]
        tb = pos_flatten_regress_targets[:, [1, 3]]
        return torch.sqrt((lr.min(dim=-1)[0] / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0] / tb.max(dim=-1)[0]))

    def distance2bbox(self, points, distance, max_shape=None):
        """fcos.py:217-238"""
        return LFD.distance2bbox(self, points, distance, max_shape)

    # ------------------------------------------------------------------ loss
    def get_loss(self, predict_outputs, annotation_batch, *args):
        """fcos.py:240-317"""
        pred_cls, pred_reg, pred_ctr = predict_outputs
        dev = pred_cls.device
        gt_b = [torch.from_numpy(b) for b, _ in annotation_batch]
        gt_l = [torch.from_numpy(l) for _, l in annotation_batch]
        pts_list = self.generate_point_coordinates(self._head_indexes_to_feature_map_sizes)
        cls_t, reg_t = self.annotation_to_target(pts_list, gt_b, gt_l)
        N = pred_cls.shape[0]
        fc, ct, pos = self._flatten_classification(pred_cls, cls_t, dev)
        fr = pred_reg.reshape(-1, 4)
        fctr = pred_ctr.reshape(-1)
        rt = reg_t.reshape(-1, 4).to(dev)
        allp = torch.cat(pts_list, dim=0).repeat(N, 1).to(dev)
        num_pos = pos.nelement()
        cls_loss = self._classification_loss_func(fc, ct, avg_factor=num_pos + N)
        pr, pc = fr[pos], fctr[pos]
        if num_pos > 0:
            prt = rt[pos]
            ctr_t = self.centerness_target(prt)
            pp = allp[pos]
            reg_loss = self._regression_loss_func(self.distance2bbox(pp, pr), self.distance2bbox(pp, prt), weight=ctr_t,
                                                  avg_factor=ctr_t.sum())
            ctr_loss = self._centerness_loss_func(pc, ctr_t)
        else:
            reg_loss = pr.sum()
            ctr_loss = pc.sum()
        loss = cls_loss + reg_loss + ctr_loss
        return dict(loss=loss, loss_values=dict(loss=loss.item(), classification_loss=cls_loss.item(),
                                                regression_loss=reg_loss.item(), centerness_loss=ctr_loss.item()))

    # ------------------------------------------------------------------ post-processing
    def _detect_desc(self, score_thr, iou_thr, class_agnostic, max_candidates=None):
        sizes = [self._head_indexes_to_feature_map_sizes[i] for i in range(self._num_levels)]
        P = sum(h * w for h, w in sizes)
        cap = max(1, min(max_candidates or self.max_candidates, P * self._num_classes))
        ranges = [(float(lo), float(min(hi, 3e38))) for lo, hi in self._regress_ranges]   # unused by decode mode 3
        return ops.make_detect_desc(sizes, self._point_strides, ranges, self._num_classes, self._num_classes, 0, 3,
                                    class_agnostic, cap, score_thr, iou_thr)

    def detect(self, predict_outputs, meta, score_thr=None, iou_thr=None, class_agnostic=None, max_candidates=None):
        """Whole-batch device post-processing (no host sync): ops.DetectOutputs.  meta [N,3] = clamp width, clamp height,
        resize scale on the device."""
        cls, reg, ctr = predict_outputs
        score_thr = self._classification_threshold if score_thr is None else score_thr
        iou_thr = self._nms_cfg.get('iou_thr', 0.5) if iou_thr is None else iou_thr
        agn = self._nms_cfg.get('class_agnostic', False) if class_agnostic is None else class_agnostic
        desc = self._detect_desc(score_thr, iou_thr, agn, max_candidates)
        return ops.detect_batched_ex(desc, cls, reg, meta, centerness=ctr, pre_nms_limit=self._pre_nms_bbox_limit,
                                     post_nms_limit=self._post_nms_bbox_limit)

    def get_results(self, predict_outputs, *args):
        """fcos.py:319-354: per image a list of [label, score, x1, y1, w, h] rows ([] when nothing survives)"""
        return _results_from_detect(self, predict_outputs, args[0])


class FCOSv1(FCOS):
    """lfd/model/fcos.py:452-900 -- "one point may predict several classes": the classification target of a point is a
    row of C binary labels (0 = this class is present at the point, 1 = background; :611-616) instead of one label, the
    logits are flattened to [N*P*C, 1] for a one-class focal loss (:711), a point is positive when any of its classes is
    (:715-716).  The regression / centerness targets, the network and get_results are FCOS's."""

    def _empty_labels(self, gt_labels, P):
        # (the reference returns FCOS's [P] vector of C here, :570-572, which its own get_loss cannot reshape to [*, C]
        # unless C == 1; an all-background row per point is what that branch means)
        return gt_labels.new_ones((P, self._num_classes))

    def _labels(self, gt_labels, valid, best, best_area):
        t = gt_labels.new_ones((valid.size(0), self._num_classes))
        rows, cols = torch.where(valid)
        t[rows, gt_labels[cols]] = 0
        return t

    def _flatten_classification(self, pred_cls, cls_t, dev):
        ct = cls_t.reshape(-1, self._num_classes)
        pos = (ct == 0).sum(dim=1).nonzero().reshape(-1).to(dev)
        return pred_cls.reshape(-1, 1), ct.reshape(-1).to(dev), pos


def _results_from_detect(model, predict_outputs, meta_batch):
    cls = predict_outputs[0]
    meta = torch.tensor([[float(m['resized_width']), float(m['resized_height']), float(m['resize_scale'])]
                         for m in meta_batch], dtype=torch.float32, device=cls.device)
    out = model.detect(predict_outputs, meta)
    counts = out.counts.cpu()
    if bool((counts[:, 2] != 0).any()):       # candidate capacity exceeded: rerun with the exact need
        out = model.detect(predict_outputs, meta, max_candidates=int(counts[:, 3].max()))
        counts = out.counts.cpu()
    return [LFD._pack(out.dets[i, :int(counts[i, 1])], out.labels[i, :int(counts[i, 1])]) for i in range(cls.size(0))]
