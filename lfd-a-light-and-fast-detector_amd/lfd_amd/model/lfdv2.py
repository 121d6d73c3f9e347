"""LFDv2 meta-architecture -- host-side mirror of lfd/model/lfdv2.py:134-815 (class LFDv2; the experimental LFDv2_
:963-1651 and the TensorRT method :817-960 are not mirrored).

LFDv2 is LFD with (a) a different training target: a point's classification target for a class is a centerness-like score
in [0,1] -- sqrt(min(l,r)/max(l,r) * min(t,b)/max(t,b)), 1 inside the box's core zone -- scaled by a linear "relaxation"
across the gray band of the level's range instead of LFD's ignore band (:278-418); (b) the `sqrt` range-assign mode;
(c) get_results with a per-level pre-NMS top-k and a post-NMS cap (:593-669).  Everything else (forward contract, decode
modes, predict_for_single_image :704-815 == LFD's) is inherited from ..lfd.LFD.

Where the arithmetic runs: forward on the fused LFD plan when neck / head are what that plan covers (SimpleNeck + 1x1
LFDHead), otherwise layer by layer on the same kernels (engine_sibling: FPN / SimpleFPN necks, 3x3 head convs);
get_results as one lfd_detect_batched_ex pass per batch; targets with the reference's tensor algebra on the host (where
the reference computes them), losses on the HIP loss kernels.
"""
import torch

from .. import ops
from .lfd import LFD
from .fcos import _results_from_detect

__all__ = ['LFDv2', 'bbox_overlaps']

from .losses.iou_loss import bbox_overlaps  # noqa: E402,F401  (lfdv2.py:15-131 re-defines it at module level)


class LFDv2(LFD):

    _ASSIGN_MODES = ('longer', 'shorter', 'sqrt', 'dist')      # lfdv2.py:158

    def __init__(self, backbone=None, neck=None, head=None, num_classes=80,
                 regression_ranges=((0, 64), (64, 128), (128, 256), (256, 512), (512, 1024)),
                 gray_range_factors=(0.9, 1.1), range_assign_mode='longer', point_strides=(8, 16, 32, 64, 128),
                 classification_loss_func=None, regression_loss_func=None, distance_to_bbox_mode='exp',
                 enable_classification_weight=False, enable_regression_weight=False, classification_threshold=0.05,
                 nms_threshold=0.5, pre_nms_bbox_limit=1000, post_nms_bbox_limit=100):
        super().__init__(backbone=backbone, neck=neck, head=head, num_classes=num_classes,
                         regression_ranges=regression_ranges, gray_range_factors=gray_range_factors,
                         range_assign_mode=range_assign_mode, point_strides=point_strides,
                         classification_loss_func=classification_loss_func, regression_loss_func=regression_loss_func,
                         distance_to_bbox_mode=distance_to_bbox_mode,
                         enable_classification_weight=enable_classification_weight,
                         enable_regression_weight=enable_regression_weight,
                         classification_threshold=classification_threshold, nms_threshold=nms_threshold)
        self._pre_nms_bbox_limit = pre_nms_bbox_limit
        self._post_nms_bbox_limit = post_nms_bbox_limit

    # forward: LFD's (fused plan when neck / head are what it covers, else the layer engine; training through autograd)
    def detect_resident(self, x, meta, score_thr=None, iou_thr=None, class_agnostic=None, max_candidates=None, slot=0):
        # the whole-step graph of LFD fuses LFD's candidate rule into the head; LFDv2's adds a per-level top-k
        return self.detect(self.forward_resident(x, slot), meta, score_thr, iou_thr, class_agnostic, max_candidates)

    # ------------------------------------------------------------------ targets
    def annotation_to_target(self, all_point_coordinates_list, gt_bboxes_list, gt_labels_list, *args):
        """lfdv2.py:232-276 (host tensors, like the reference)"""
        n_per = [p.size(0) for p in all_point_coordinates_list]
        pts = torch.cat(all_point_coordinates_list, 0)
        rr = torch.cat([pts.new_tensor(self._regression_ranges[i])[None].expand(n_per[i], 2) for i in range(self._num_heads)])
        gr = torch.cat([pts.new_tensor(self._gray_ranges[i])[None].expand(n_per[i], 2) for i in range(self._num_heads)])
        st = torch.cat([pts.new_tensor(self._point_strides[i]).expand(n_per[i]) for i in range(self._num_heads)])
        cls_t, reg_t = [], []
        for b, l in zip(gt_bboxes_list, gt_labels_list):
            c, r = self._generate_target_for_single_image(b, l, pts.to(b.device), rr.to(b.device), gr.to(b.device),
                                                          st.to(b.device))
            cls_t.append(c)
            reg_t.append(r)
        return torch.stack(cls_t, 0), torch.stack(reg_t, 0)

    def _generate_target_for_single_image(self, gt_bboxes, gt_labels, points, reg_ranges, gray_ranges, strides):
        """lfdv2.py:278-418, same expression order (fp32)"""
        assert gt_bboxes.size(0) == gt_labels.size(0)
        P, G = points.size(0), gt_bboxes.size(0)
        cls_t = gt_bboxes.new_full((P, self._num_classes), 0)
        reg_t = gt_bboxes.new_zeros((P, 4))
        if G == 0:
            return cls_t, reg_t
        gb = gt_bboxes[None].expand(P, G, 4)
        gl = gt_labels[None].expand(P, G)
        rr = reg_ranges[:, None, :].expand(P, G, 2)
        gr = gray_ranges[:, None, :].expand(P, G, 2)
        px = points[:, 0][:, None].expand(P, G)
        py = points[:, 1][:, None].expand(P, G)
        cx = gb[..., 0] + gb[..., 2] / 2.
        cy = gb[..., 1] + gb[..., 3] / 2.
        delta = torch.stack((px - gb[..., 0], py - gb[..., 1],
                             (gb[..., 0] + gb[..., 2] - 1) - px, (gb[..., 1] + gb[..., 3] - 1) - py), dim=-1)
        hit = delta.min(dim=-1)[0] >= 0
        # centerness-like score of a point inside a box (0 outside), 1 in the stride-sized core zone around the centre
        fd = delta * hit[..., None].expand((P, G, 4))
        lr, tb = fd[..., [0, 2]], fd[..., [1, 3]]
        score = ((lr.min(dim=-1)[0]).clamp(min=0.0) / (lr.max(dim=-1)[0]).clamp(min=0.01)) * \
                ((tb.min(dim=-1)[0]).clamp(min=0.0) / (tb.max(dim=-1)[0]).clamp(min=0.01))
        score = torch.sqrt(score)
        half = strides[:, None].expand((P, G)) / 2
        core = (px >= cx - half) & (px <= cx + half) & (py >= cy - half) & (py <= cy + half) & hit
        score = score * (~core) + core
        mode = self._range_assign_mode
        if mode == 'longer':
            measure = torch.max(gb[..., 2], gb[..., 3])
        elif mode == 'shorter':
            measure = torch.min(gb[..., 2], gb[..., 3])
        elif mode == 'sqrt':
            measure = torch.sqrt(gb[..., 2] * gb[..., 3])
        elif mode == 'dist':
            measure = delta.max(dim=-1)[0]
        else:
            raise ValueError('Unsupported range assign mode!')
        if self._regression_loss_type == 'independent':
            delta = delta / rr[..., 1, None]
        # relaxation across the gray band: linear ramp up to the range, 1 inside, linear ramp down after it
        left = (measure - gr[..., 0]) / (rr[..., 0] - gr[..., 0]).clamp(min=0.01)
        left_on = (gr[..., 0] <= measure) & (measure < rr[..., 0])
        inside = (rr[..., 0] <= measure) & (measure <= rr[..., 1])
        right = (gr[..., 1] - measure) / (gr[..., 1] - rr[..., 1]).clamp(min=0.01)
        right_on = (rr[..., 1] < measure) & (measure <= gr[..., 1])
        score = score * (left * left_on + inside + right * right_on)
        positive = score > 0
        sscore, sidx = score.sort(dim=1)           # ascending: the largest score of a class is written last
        rows = torch.arange(P, device=points.device)[:, None].expand(P, G)
        sl, spos = gl[rows, sidx], positive[rows, sidx]
        i1, i2 = torch.where(spos)
        cls_t[i1, sl[i1, i2]] = sscore[i1, i2]
        sel = sscore.max(dim=1)[1]
        reg_t = delta[rows, sidx][torch.arange(P, device=points.device), sel]
        return cls_t, reg_t

    def _fused_loss_supported(self, pred_cls):
        return False        # the device target kernel implements LFD's assignment, not LFDv2's

    def get_loss(self, predict_outputs, annotation_batch, *args):
        """lfdv2.py:443-554: same reduction as LFD.get_loss over LFDv2's targets (built on the host, as in the reference)"""
        pred_cls, pred_reg = predict_outputs
        gt_b = [torch.as_tensor(b) for b, _ in annotation_batch]
        gt_l = [torch.as_tensor(l) for _, l in annotation_batch]
        pts_list = self.generate_point_coordinates(self._head_indexes_to_feature_map_sizes)
        cls_t, reg_t = self.annotation_to_target(pts_list, gt_b, gt_l)
        return self._loss_from_targets(pred_cls, pred_reg, cls_t, reg_t, pts_list)

    # ------------------------------------------------------------------ post-processing
    def detect(self, predict_outputs, meta, score_thr=None, iou_thr=None, class_agnostic=None, max_candidates=None):
        cls, reg = predict_outputs[0], predict_outputs[1]
        score_thr = self._classification_threshold if score_thr is None else score_thr
        iou_thr = self._nms_cfg.get('iou_thr', 0.5) if iou_thr is None else iou_thr
        agn = self._nms_cfg.get('class_agnostic', False) if class_agnostic is None else class_agnostic
        desc, _ = self._detect_desc(score_thr, iou_thr, agn, max_candidates)
        return ops.detect_batched_ex(desc, cls, reg, meta, centerness=None, pre_nms_limit=self._pre_nms_bbox_limit,
                                     post_nms_limit=self._post_nms_bbox_limit)

    def get_results(self, predict_outputs, *args):
        """lfdv2.py:556-591"""
        return _results_from_detect(self, predict_outputs, args[0])

    def _detect_with_retry(self, cls, reg, meta, score_thr, iou_thr, agn):
        """predict_for_single_image (lfdv2.py:704-815) thresholds every point and passes max_num=-1: LFD's rule"""
        out = LFD.detect(self, (cls, reg), meta, score_thr, iou_thr, agn)
        counts = out.counts.cpu()
        if bool((counts[:, 2] != 0).any()):
            out = LFD.detect(self, (cls, reg), meta, score_thr, iou_thr, agn, max_candidates=int(counts[:, 3].max()))
            counts = out.counts.cpu()
        return out, counts
