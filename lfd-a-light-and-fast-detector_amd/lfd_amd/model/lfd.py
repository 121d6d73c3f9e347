"""LFD meta-architecture -- host-side mirror of lfd/model/lfd.py:15-655 (the TensorRT method
:657-800 is out of scope).  Same constructor kwargs, `forward(x) -> (cls[N,P,C'], reg[N,P,4])`
fp32, `head_indexes_to_feature_map_sizes`, `get_results`, `predict_for_single_image`,
`get_loss` contracts; the arithmetic runs in liblfd_hip.so:

  eval-mode forward      -> engine.lfd_forward (stem / conv / head kernels, csrc/*.hip)
  get_results / predict  -> ONE fused device pass for the whole batch (decode + threshold +
                            per-class NMS, csrc/postproc.hip) instead of the reference's
                            per-image Python loop with ~20 tiny ATen kernels per level and a
                            blocking D2H of the NMS mask per image (lfd.py:412-431, nms_kernel.cu:105-111)
  get_loss               -> device target assignment (csrc/targets.hip) + the three-launch fused loss
                            (csrc/getloss.hip); other loss modules take the op-by-op path on the HIP loss kernels

Train-mode forward (`self.training`): for the shipped configurations the whole network runs forward and
backward on the hand-written kernels as one autograd node (train_engine.NetworkTrainFunction, csrc/train.hip;
per-replica BatchNorm batch statistics exactly like the reference).  `LFD_HIP_TRAIN=0`, or a module
configuration train_engine does not cover (none of the six named configs), routes the same parameters through
PyTorch-ROCm autograd instead -- a documented training-only fallback; inference has none.
"""
import os

import numpy
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, engine_p32, engine_sibling, ops, parallel, train_engine
from .utils import multiclass_nms  # noqa: F401  (kept importable like the reference module)

__all__ = ['LFD']

_UNION = ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss')


class _FusedLossFunction(torch.autograd.Function):
    """get_loss as lfd_get_loss_{sums,finalize,bwd}_f32 (csrc/getloss.hip): -> [classification_loss,
    regression_loss, loss]; backward writes the dense prediction gradients in one launch."""

    @staticmethod
    def forward(ctx, pred_cls, pred_reg, cls_t, reg_t, desc, reduce_sums, rank_scale):
        fin = ops.get_loss_forward(desc, pred_cls, pred_reg, cls_t, reg_t, reduce_sums, rank_scale)
        ctx.desc = desc
        ctx.save_for_backward(pred_cls, pred_reg, cls_t, reg_t, fin)
        return fin[:3].clone()

    @staticmethod
    def backward(ctx, g):
        pred_cls, pred_reg, cls_t, reg_t, fin = ctx.saved_tensors
        gc, gr = ops.get_loss_backward(ctx.desc, pred_cls, pred_reg, cls_t, reg_t, fin, g)
        return gc.view_as(pred_cls).to(pred_cls.dtype), gr.view_as(pred_reg).to(pred_reg.dtype), None, None, None, None, None


class LFD(nn.Module):

    _ASSIGN_MODES = ('longer', 'shorter', 'dist')       # lfd.py:35

    def __init__(self, backbone=None, neck=None, head=None, num_classes=80,
                 regression_ranges=((0, 64), (64, 128), (128, 256), (256, 512), (512, 1024)),
                 gray_range_factors=(0.9, 1.1), range_assign_mode='dist', point_strides=(8, 16, 32, 64, 128),
                 classification_loss_func=None, regression_loss_func=None, distance_to_bbox_mode='exp',
                 enable_classification_weight=False, enable_regression_weight=False,
                 classification_threshold=0.05, nms_threshold=0.4):
        super().__init__()
        assert len(regression_ranges) == len(point_strides)
        assert range_assign_mode in self._ASSIGN_MODES
        assert distance_to_bbox_mode in ['exp', 'sigmoid']
        self._backbone, self._neck, self._head = backbone, neck, head
        self._num_classes = num_classes
        self._regression_ranges = regression_ranges
        self._range_assign_mode = range_assign_mode
        if range_assign_mode in ('shorter', 'sqrt'):
            assert type(regression_loss_func).__name__ in _UNION
            assert distance_to_bbox_mode == 'exp'
        self._gray_range_factors = (min(gray_range_factors), max(gray_range_factors))
        self._gray_ranges = [(int(lo * self._gray_range_factors[0]), int(up * self._gray_range_factors[1]))
                             for (lo, up) in regression_ranges]
        self._num_heads = len(point_strides)
        self._point_strides = point_strides
        if classification_loss_func is not None:
            assert type(classification_loss_func).__name__ in ['BCEWithLogitsLoss', 'FocalLoss', 'CrossEntropyLoss',
                                                               'QualityFocalLoss']
        self._classification_loss_func = classification_loss_func
        if regression_loss_func is not None:
            name = type(regression_loss_func).__name__
            assert name in ['SmoothL1Loss', 'MSELoss'] + list(_UNION)
            self._regression_loss_type = 'independent' if name in ['SmoothL1Loss', 'MSELoss'] else 'union'
        self._regression_loss_func = regression_loss_func
        self._distance_to_bbox_mode = distance_to_bbox_mode
        self._enable_classification_weight = enable_classification_weight
        self._enable_regression_weight = enable_regression_weight
        self._classification_threshold = classification_threshold
        self._nms_cfg = dict(type='nms', iou_thr=nms_threshold)
        self._head_indexes_to_feature_map_sizes = dict()
        self.max_candidates = 8192   # per-image capacity of the fused post-processing pass
        self.use_graph = False       # capture the forward into a HIP graph per input shape
        self._precision = 'fp16'
        self.register_load_state_dict_post_hook(LFD._forget_cached_tensors)

    PRECISIONS = ('fp16', 'fp32_storage')

    @property
    def precision(self):
        """Numeric mode of the eval-mode forward (training is unaffected):
        'fp16'          fp16 MFMA operands and fp16 inter-layer storage, fp32 accumulation -- the fused stem / block / head
                        kernels of engine.py; sigma(cls) / sigma(reg) within 1.3e-3 .. 2.3e-3 of the fp32 reference;
        'fp32_storage'  every inter-layer tensor as fp16 hi + 2^-11 lo planes (the bytes of fp32), weights split the same
                        way, three MFMAs per k-step, fp32 accumulation and epilogues (engine_p2.py, csrc/planes*.hip;
                        engine_p32.py / csrc/precise.hip for layer shapes without a plane kernel); raw logits within 1e-4
                        of the fp32 reference (north_star: "tensors within 1e-3"; measured 8e-6), ~3.5x the time."""
        return self.__dict__.get('_precision', 'fp16')

    @precision.setter
    def precision(self, value):
        if value not in self.PRECISIONS:
            raise ValueError('precision must be one of %s' % (self.PRECISIONS,))
        if value != self.precision:
            self.__dict__.get('_step_graphs', {}).clear()
            self.__dict__['_step_graphs_ver'] = None
        self._precision = value

    @staticmethod
    def _forget_cached_tensors(module, incompatible_keys=None):
        # load_state_dict (assign=True swaps the parameter objects): the cached tensor list of engine.version_sum is stale
        module.__dict__.pop('_lfd_tensors', None)
        module.__dict__['_step_graphs_ver'] = None

    def __setattr__(self, name, value):
        if name in ('_backbone', '_neck', '_head'):      # a replaced submodule: new tensors, maybe a new plan family
            self.__dict__.pop('_lfd_tensors', None)
            self.__dict__.pop('_fused_ok', None)
            self.__dict__['_step_graphs_ver'] = None
        super().__setattr__(name, value)

    def _apply(self, fn, recurse=True):
        # .to() / .cuda() / .half(): parameter storage is replaced -> forget the cached tensor list and captured graphs
        self.__dict__.pop('_lfd_tensors', None)
        self.__dict__.pop('_fused_ok', None)
        self.__dict__['_step_graphs_ver'] = None
        self.__dict__.get('_step_graphs', {}).clear()
        return super()._apply(fn, recurse)

    @property
    def head_indexes_to_feature_map_sizes(self):
        return self._head_indexes_to_feature_map_sizes

    # ------------------------------------------------------------------ forward
    def _is_ce(self):
        return type(self._classification_loss_func).__name__ == 'CrossEntropyLoss'

    def _fused_plan_ok(self, device):
        """the fused plan of engine.py (stem / block / 3-pass head kernels) covers SimpleNeck + an LFDHead of 1x1 convs --
        every configuration the reference ships; other module combinations its constructors allow (FPN / SimpleFPN necks,
        3x3 head convs, LFDHeadV1) run layer by layer on the same kernels (engine_sibling)"""
        ok = self.__dict__.get('_fused_ok')
        if ok is None:
            ok = type(self._neck).__name__ == 'SimpleNeck' and type(self._head).__name__ == 'LFDHead' \
                and self._head._conv_kernel_size == 1
            if ok:
                try:
                    engine.get_plan(self, self._backbone, self._neck, self._head, device)
                except engine.Unsupported:
                    ok = False       # only "this module tree is not covered" selects the layer engine; an out-of-memory or
                                     # a library failure propagates instead of silently (and permanently) slowing the model
            self.__dict__['_fused_ok'] = ok
        return ok

    def forward(self, x):
        """lfd.py:511-542.  eval mode: HIP engine.  Returns fresh fp32 tensors (clone of the
        engine's resident output buffers) and records (h, w) per level (:532)."""
        if self.training:
            return self._forward_train(x)
        cls, reg = self.forward_resident(x)
        return cls.clone(), reg.clone()

    def forward_resident(self, x, slot=0):
        """Fast path: same as forward() in eval mode but returns the engine-owned output buffers
        (no copy, valid until the next forward of the same input shape and `slot`)."""
        if self.precision == 'fp32_storage':
            cls, reg, sizes = engine_p32.lfd_forward(self, x, use_graph=self.use_graph, slot=slot)
        elif x.is_cuda and not self._fused_plan_ok(x.device):
            cls, reg, _, sizes = engine_sibling.sibling_forward(self, x, use_graph=self.use_graph)
        else:
            cls, reg, sizes = engine.lfd_forward(self, x, use_graph=self.use_graph, slot=slot)
        for i, hw in enumerate(sizes):
            self._head_indexes_to_feature_map_sizes[i] = hw
        return cls, reg

    def detect_resident(self, x, meta, score_thr=None, iou_thr=None, class_agnostic=None, max_candidates=None, slot=0):
        """Whole inference step for frames resident in device memory: forward + decode + threshold + NMS, results on the
        device (ops.DetectOutputs, overwritten by the next call with the same buffers).  With `use_graph` the complete
        step -- not only the forward -- is one HIP graph per (frame buffer, meta buffer, thresholds, slot): one host call
        per step, no launch gap between the head and the post-processing kernels.  `slot` selects an independent set of
        activation / output / workspace buffers: steps of DIFFERENT slots may be enqueued on different HIP streams and
        overlap on the device (the small-map stages, the post-processing kernels and the launch gaps of one batch leave most
        CUs idle -- a second batch in flight fills them: 0.75 -> 0.61 ms per batch of 8 at depth 2, tools/ab_pipeline.py)."""
        precise = self.precision == 'fp32_storage'
        if not self.use_graph or (not precise and not self._fused_plan_ok(x.device)):
            return self.detect(self.forward_resident(x, slot), meta, score_thr, iou_thr, class_agnostic, max_candidates)
        score_thr = self._classification_threshold if score_thr is None else score_thr
        iou_thr = self._nms_cfg.get('iou_thr', 0.5) if iou_thr is None else iou_thr
        agn = self._nms_cfg.get('class_agnostic', False) if class_agnostic is None else class_agnostic
        cache = self.__dict__.setdefault('_step_graphs', {})
        # the captured graph reads the packed weights of ONE EnginePlan: a parameter change (optimizer step,
        # load_state_dict, .to()) makes get_plan build a new plan, and every graph captured against the old one is
        # dropped here (replaying it would silently use the old weights -- or freed memory)
        # The full check (engine.get_plan: data_ptr + version of every tensor) costs ~0.15 ms of Python -- as much as a
        # whole bs-1 step; in-place version counters only ever increase, so their SUM changes iff some tensor was written
        # in place (optimizer step, load_state_dict, .copy_); storage moves (.to / .cuda / .half) go through _apply below.
        ver = engine.version_sum(self)
        if self.__dict__.get('_step_graphs_ver') != ver:
            plan = engine_p32.get_plan(self, x.device) if precise else \
                engine.get_plan(self, self._backbone, self._neck, self._head, x.device)
            if self.__dict__.get('_step_graphs_plan') is not plan:
                cache.clear()
                self.__dict__['_step_graphs_plan'] = plan
            self.__dict__['_step_graphs_ver'] = ver
        key = (x.data_ptr(), tuple(x.shape), x.dtype, meta.data_ptr(), float(score_thr), float(iou_thr), bool(agn), max_candidates, slot)
        ent = cache.get(key)
        if ent is None:
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            self.use_graph = False
            try:
                with torch.cuda.device(x.device):
                    out = self.detect(self.forward_resident(x, slot), meta, score_thr, iou_thr, agn, max_candidates)   # warm-up, sizes buffers
                    torch.cuda.synchronize()
                    desc, _ = self._detect_desc(score_thr, iou_thr, agn, max_candidates)
                    # single-class sigmoid models: the head's output pass thresholds, decodes and appends the candidates
                    # itself (lfd_head_forward_decode_f16), logits never reach HBM, 3 post-processing launches instead of 5
                    ops.detect_workspace_reset(desc, x.size(0), out)
                    fused = (not precise) and engine.lfd_forward_detect(self, x, desc, meta, out, slot)   # warm-up of the fused kernels
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        if fused:
                            engine.lfd_forward_detect(self, x, desc, meta, out, slot)
                        else:
                            cls, reg = self.forward_resident(x, slot)
                            ops.detect_batched(desc, cls, reg, meta, out=out)
            finally:
                self.use_graph = True
            ent = (g, out, x, meta)          # keep the captured buffers alive
            cache[key] = ent
        ent[0].replay()
        return ent[1]

    def _forward_train(self, x):
        """Train-mode forward (batch-statistic BatchNorm, per-image GroupNorm, autograd graph).  For the shipped
        configurations the WHOLE network -- backbone, neck, head towers, output convs -- runs forward and backward on the
        hand-written kernels as one autograd node (train_engine.NetworkTrainFunction); a supported backbone under an
        unsupported neck / head runs as its own node with PyTorch-ROCm modules behind it.  LFD_HIP_TRAIN=0 or an
        unsupported backbone: everything through PyTorch-ROCm autograd (same parameters, same semantics)."""
        neck, head = self._neck, self._head
        hip = x.is_cuda and os.environ.get('LFD_HIP_TRAIN', '1') != '0'
        if hip and train_engine.network_supported(self):
            cls, reg, sizes = train_engine.network_train_forward(self, x)
            for i, sz in enumerate(sizes):
                self._head_indexes_to_feature_map_sizes[i] = sz
            return cls, reg
        if hip and train_engine.supported(self._backbone):
            feats = list(train_engine.backbone_train_forward(self._backbone, x))
        else:
            feats = self._backbone_train_torch(x)
        if type(neck).__name__ == 'SimpleNeck':
            feats = [getattr(neck, 'neck%d' % i)(f) for i, f in enumerate(feats)]
        else:
            feats = neck(feats)                    # FPN / SimpleFPN: autograd over their PyTorch-ROCm children
        cls_l, reg_l = [], []
        for i, t in enumerate(feats):
            t = getattr(head, 'head%d_merge_path' % i)(t)
            c = getattr(head, 'head%d_classification_path' % i)(t)
            r = getattr(head, 'head%d_regression_path' % i)(t)
            if hasattr(head, '_classifiers'):      # LFDHeadV1: per-level output convs outside the (shared) towers
                c, r = head._classifiers[i](c), head._regressors[i](r)
            if head._regression_loss_type in _UNION:
                r = head._scales[i](r)
            self._head_indexes_to_feature_map_sizes[i] = (c.shape[2], c.shape[3])
            cls_l.append(c.permute(0, 2, 3, 1).reshape(c.shape[0], -1, c.shape[1]))
            reg_l.append(r.permute(0, 2, 3, 1).reshape(r.shape[0], -1, 4))
        return torch.cat(cls_l, 1), torch.cat(reg_l, 1)

    def _backbone_train_torch(self, x):
        bb = self._backbone
        y = bb._stem(x)
        feats = []
        taps = [tuple(t) for t in bb._out_indices]
        for i, nblk in enumerate(bb._body_architecture):
            for j in range(nblk):
                blk = getattr(bb, 'stage%d' % i)[j]
                ident = y if blk._downsample is None else blk._downsample(y)
                o = y
                for ci in range(1, blk.num_convs + 1):
                    o = getattr(blk, '_conv%d' % ci)(o)
                    if blk._norm_cfg is not None:
                        o = getattr(blk, '_norm%d' % ci)(o)
                    if ci < blk.num_convs:
                        o = F.relu(o)
                y = F.relu(o + ident)
                if (i, j) in taps:
                    feats.append(y)
        return feats

    # ------------------------------------------------------------------ point grid / targets
    def generate_point_coordinates(self, feature_map_sizes):
        """lfd.py:84-107: x = j*stride, y = i*stride (no half-stride offset), row-major, int64 CPU."""
        assert len(feature_map_sizes) == len(self._point_strides)
        out = []
        for i, s in enumerate(self._point_strides):
            h, w = feature_map_sizes[i]
            ys, xs = torch.meshgrid(torch.arange(0, h * s, s), torch.arange(0, w * s, s), indexing='ij')
            out.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), dim=-1))
        return out

    def annotation_to_target(self, all_point_coordinates_list, gt_bboxes_list, gt_labels_list, *args):
        """lfd.py:109-153 -> ([N,P,C] cls targets, [N,P,4] reg targets), computed on the device the
        annotations live on (the reference does it on the CPU with [P,G] broadcasts)."""
        dev = gt_bboxes_list[0].device if len(gt_bboxes_list) else torch.device('cpu')
        if dev.type == 'cuda':
            # device path: one launch of lfd_assign_targets_f32 for the whole batch (csrc/targets.hip); the point
            # grid is regenerated on the fly from the recorded feature-map sizes (same values as the list passed in)
            sizes = [self._head_indexes_to_feature_map_sizes[i] for i in range(self._num_heads)]
            assert [p.size(0) for p in all_point_coordinates_list] == [h * w for h, w in sizes]
            return ops.assign_targets(sizes, self._point_strides, self._regression_ranges, self._gray_ranges,
                                      self._num_classes, self._range_assign_mode,
                                      self._regression_loss_type == 'independent', gt_bboxes_list, gt_labels_list)
        # CPU tensors (where the reference itself runs this function): the reference's tensor algebra
        pts = torch.cat(all_point_coordinates_list, 0).to(dev)
        n_per = [p.size(0) for p in all_point_coordinates_list]
        rr = torch.cat([torch.tensor(self._regression_ranges[i], dtype=torch.int64)[None].expand(n_per[i], 2)
                        for i in range(self._num_heads)]).to(dev)
        gr = torch.cat([torch.tensor(self._gray_ranges[i], dtype=torch.int64)[None].expand(n_per[i], 2)
                        for i in range(self._num_heads)]).to(dev)
        st = torch.cat([torch.full((n_per[i],), self._point_strides[i], dtype=torch.int64)
                        for i in range(self._num_heads)]).to(dev)
        cls_t, reg_t = [], []
        for b, l in zip(gt_bboxes_list, gt_labels_list):
            c, r = self._generate_target_for_single_image(b, l, pts, rr, gr, st)
            cls_t.append(c)
            reg_t.append(r)
        return torch.stack(cls_t, 0), torch.stack(reg_t, 0)

    def _generate_target_for_single_image(self, gt_bboxes, gt_labels, points, reg_ranges, gray_ranges, strides):
        """lfd.py:155-259, same expression order (fp32)."""
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
        st = strides[:, None]

        def axis_score(d):
            v = d / (st / 2.)
            v = v * (v >= 1) + (v < 1)
            return torch.sqrt(1. / v)

        score = axis_score(torch.abs(px - cx)) * axis_score(torch.abs(py - cy))
        delta = torch.stack((px - gb[..., 0], py - gb[..., 1],
                             (gb[..., 0] + gb[..., 2] - 1) - px, (gb[..., 1] + gb[..., 3] - 1) - py), dim=-1)
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
        hit = delta.min(dim=-1)[0] >= 0
        green = (rr[..., 0] <= measure) & (measure <= rr[..., 1]) & hit
        gray = (((gr[..., 0] <= measure) & (measure < rr[..., 0])) |
                ((rr[..., 1] < measure) & (measure <= gr[..., 1]))) & hit
        sscore, sidx = score.sort(dim=1)   # ascending: the largest score is written last (lfd.py:230-235)
        rows = torch.arange(P, device=points.device)[:, None].expand(P, G)
        sl, sgreen, sgray = gl[rows, sidx], green[rows, sidx], gray[rows, sidx]
        i1, i2 = torch.where(sgreen)
        cls_t[i1, sl[i1, i2]] = sscore[i1, i2]
        i3, i4 = torch.where(sgray)
        cls_t[i3, sl[i3, i4]] = -1
        _, sel = (sscore * (sgreen & ~sgray)).max(dim=1)
        reg_t = delta[rows, sidx][torch.arange(P, device=points.device), sel]
        return cls_t, reg_t

    def distance2bbox(self, points, distance, max_shape=None):
        """lfd.py:261-282."""
        x1 = points[:, 0] - distance[:, 0]
        y1 = points[:, 1] - distance[:, 1]
        x2 = points[:, 0] + distance[:, 2]
        y2 = points[:, 1] + distance[:, 3]
        if max_shape is not None:
            x1 = x1.clamp(min=0, max=max_shape[1])
            y1 = y1.clamp(min=0, max=max_shape[0])
            x2 = x2.clamp(min=0, max=max_shape[1])
            y2 = y2.clamp(min=0, max=max_shape[0])
        return torch.stack([x1, y1, x2, y2], -1)

    # ------------------------------------------------------------------ loss
    def get_loss(self, predict_outputs, annotation_batch, *args):
        """lfd.py:284-395.  Returns {'loss': Tensor, 'loss_values': {...floats}}."""
        pred_cls, pred_reg = predict_outputs
        dev = pred_cls.device
        if dev.type == 'cuda' and self._fused_loss_supported(pred_cls):
            # device path end to end: annotations concatenated on the host and uploaded once, targets by
            # lfd_assign_targets_f32, loss by the fused get_loss kernels; no per-image tensors, one host sync
            sizes = [self._head_indexes_to_feature_map_sizes[i] for i in range(self._num_heads)]
            cls_t, reg_t = ops.assign_targets_from_host(sizes, self._point_strides, self._regression_ranges,
                                                        self._gray_ranges, self._num_classes, self._range_assign_mode,
                                                        self._regression_loss_type == 'independent', annotation_batch, dev)
            return self._get_loss_fused(pred_cls, pred_reg, cls_t, reg_t)
        gt_b, gt_l = [], []
        for bboxes_numpy, labels_numpy in annotation_batch:
            gt_b.append(torch.as_tensor(bboxes_numpy).to(dev))
            gt_l.append(torch.as_tensor(labels_numpy).to(dev))
        pts_list = self.generate_point_coordinates(self._head_indexes_to_feature_map_sizes)
        cls_t, reg_t = self.annotation_to_target(pts_list, gt_b, gt_l)
        if self._fused_loss_supported(pred_cls):
            return self._get_loss_fused(pred_cls, pred_reg, cls_t, reg_t)
        return self._loss_from_targets(pred_cls, pred_reg, cls_t, reg_t, pts_list)

    def _loss_from_targets(self, pred_cls, pred_reg, cls_t, reg_t, pts_list):
        """lfd.py:300-395 from the flattening on: op by op on the HIP loss kernels (any loss module combination)"""
        dev = pred_cls.device
        N = pred_cls.size(0)
        C = self._num_classes
        ce = self._is_ce()
        fc = pred_cls.reshape(-1, C + 1 if ce else C)
        fr = pred_reg.reshape(-1, 4)
        ct = cls_t.reshape(-1, C).to(dev)
        rt = reg_t.reshape(-1, 4).to(dev)
        green = torch.where(ct.min(dim=-1)[0] >= 0)[0]
        fc, fr, ct, rt = fc[green], fr[green], ct[green], rt[green]
        mx, mi = ct.max(dim=-1)
        pos = torch.where(mx >= 0.001)[0]
        weight = mx[pos]
        cname = type(self._classification_loss_func).__name__
        if cname in ['FocalLoss', 'CrossEntropyLoss', 'QualityFocalLoss']:
            label = mi * (mx >= 0.001) + C * (mx < 0.001)
            tgt = [label, mx] if cname == 'QualityFocalLoss' else label
        else:
            tgt = ct
        # normalisers: global-batch quantities in the reference (the loss is computed once over the gathered outputs of all
        # replicas, executor.py:198-200); under image-parallel training they are all-reduced and the rank's loss is scaled
        # by the world size so that the averaged gradients equal the reference's (same convention as the fused path)
        n_pos_g, w_sum_g, rank_scale = pos.nelement(), weight.sum(), 1.0
        if parallel.is_dist():
            t = parallel.global_count(torch.stack([weight.new_tensor(float(pos.nelement())), weight.sum().float()]))
            n_pos_g, w_sum_g, rank_scale = t[0], t[1], float(parallel.world_size())
        avg_c = w_sum_g if self._enable_classification_weight else n_pos_g + 1
        cls_loss = self._classification_loss_func(fc, tgt, avg_factor=avg_c)
        if rank_scale != 1.0:
            cls_loss = cls_loss * rank_scale
        frp, rtp = fr[pos], rt[pos]
        if pos.nelement() > 0:
            avg_r = w_sum_g if self._enable_regression_weight else n_pos_g
            w_r = weight if self._enable_regression_weight else None
            if self._regression_loss_type == 'independent':
                reg_loss = self._regression_loss_func(frp, rtp, avg_factor=avg_r, weight=w_r)
            else:
                allp = torch.cat(pts_list, 0).to(dev).repeat(N, 1)[green][pos]
                tgt_xyxy = self.distance2bbox(allp, rtp)
                if self._distance_to_bbox_mode == 'exp':
                    d = frp.float().exp()
                else:
                    rr = torch.cat([torch.tensor(self._regression_ranges[i], dtype=torch.int64)[None]
                                    .expand(pts_list[i].size(0), 2) for i in range(self._num_heads)]).to(dev)
                    rmax = rr.repeat(N, 1)[green][pos].max(dim=-1)[0]
                    d = frp.sigmoid() * rmax[..., None]
                pred_xyxy = self.distance2bbox(allp, d)
                reg_loss = self._regression_loss_func(pred_xyxy, tgt_xyxy, avg_factor=avg_r, weight=w_r)
            if rank_scale != 1.0:
                reg_loss = reg_loss * rank_scale
        else:
            reg_loss = frp.sum()
        loss = cls_loss + reg_loss
        return dict(loss=loss, loss_values=dict(loss=loss.item(), classification_loss=cls_loss.item(),
                                                regression_loss=reg_loss.item()))

    def _fused_loss_supported(self, pred_cls):
        """The three-launch device path (csrc/getloss.hip) covers what the shipped configs use: FocalLoss or
        CrossEntropyLoss + IoULoss, 'mean' reduction; other loss modules take the op-by-op path above."""
        if pred_cls.device.type != 'cuda' or os.environ.get('LFD_FUSED_LOSS', '1') == '0':
            return False
        cf, rf = self._classification_loss_func, self._regression_loss_func
        if type(cf).__name__ == 'FocalLoss':
            if not getattr(cf, 'use_sigmoid', True):
                return False
        elif type(cf).__name__ != 'CrossEntropyLoss':
            return False
        return type(rf).__name__ == 'IoULoss' and cf.reduction == 'mean' and rf.reduction == 'mean'

    def _loss_desc(self, n):
        """lfd_loss_desc_t of the fused get_loss kernels for a batch of n images at the recorded feature-map sizes"""
        cf, rf = self._classification_loss_func, self._regression_loss_func
        sizes = [self._head_indexes_to_feature_map_sizes[i] for i in range(self._num_heads)]
        return ops.make_loss_desc(n, sizes, self._point_strides, self._regression_ranges,
                                  self._num_classes, self._is_ce(), self._distance_to_bbox_mode,
                                  gamma=getattr(cf, 'gamma', 2.0), alpha=getattr(cf, 'alpha', 0.25), iou_eps=rf.eps,
                                  cls_loss_weight=cf.loss_weight, reg_loss_weight=rf.loss_weight,
                                  cls_weighted=self._enable_classification_weight,
                                  reg_weighted=self._enable_regression_weight)

    def _fused_loss_tensor(self, pred_cls, pred_reg, cls_t, reg_t):
        """-> float32[3] device tensor (classification_loss, regression_loss, loss) with the autograd graph of the fused
        kernels behind it; no host sync (a captured training iteration reads it after the replay)"""
        desc = self._loss_desc(pred_cls.size(0))
        # image-parallel training: the reference normalises by the GLOBAL-batch n_pos (loss computed once over the
        # gathered outputs, executor.py:198-200); gradients are averaged over ranks afterwards, hence the scale
        dist_on = parallel.is_dist()
        return _FusedLossFunction.apply(pred_cls, pred_reg, cls_t, reg_t, desc,
                                        parallel.global_count if dist_on else None,
                                        float(parallel.world_size()) if dist_on else 1.0)

    def _get_loss_fused(self, pred_cls, pred_reg, cls_t, reg_t):
        vals = self._fused_loss_tensor(pred_cls, pred_reg, cls_t, reg_t)
        c, r, t = vals.tolist()      # the one host sync of the step (the reference does three .item())
        return dict(loss=vals[2], loss_values=dict(loss=t, classification_loss=c, regression_loss=r))

    # ------------------------------------------------------------------ post-processing
    def _detect_desc(self, score_thr, iou_thr, class_agnostic, max_candidates=None):
        sizes = [self._head_indexes_to_feature_map_sizes[i] for i in range(self._num_heads)]
        if self._regression_loss_type == 'independent':
            mode = 2
        else:
            mode = 1 if self._distance_to_bbox_mode == 'exp' else 0
        P = sum(h * w for h, w in sizes)
        cap = max_candidates or self.max_candidates
        cap = max(1, min(cap, P * self._num_classes))
        ce = self._is_ce()
        return ops.make_detect_desc(sizes, self._point_strides, self._regression_ranges, self._num_classes,
                                    self._num_classes + (1 if ce else 0), 1 if ce else 0, mode, class_agnostic, cap,
                                    score_thr, iou_thr), P

    def detect(self, predict_outputs, meta, score_thr=None, iou_thr=None, class_agnostic=None, max_candidates=None):
        """Fused device post-processing for the whole batch, no host sync.  meta: float tensor
        [N,3] = (clamp_width, clamp_height, resize_scale) on the device.  Returns ops.DetectOutputs."""
        cls, reg = predict_outputs
        score_thr = self._classification_threshold if score_thr is None else score_thr
        iou_thr = self._nms_cfg.get('iou_thr', 0.5) if iou_thr is None else iou_thr
        agn = self._nms_cfg.get('class_agnostic', False) if class_agnostic is None else class_agnostic
        desc, _ = self._detect_desc(score_thr, iou_thr, agn, max_candidates)
        return ops.detect_batched(desc, cls, reg, meta)

    @staticmethod
    def _pack(dets, labels):
        """[x1,y1,x2,y2,score] -> [label, score, x1, y1, w, h] with the +1 (lfd.py:421-429, 646-655)."""
        if dets.size(0) == 0:
            return []
        d = dets.clone()
        d[:, 2] = d[:, 2] - d[:, 0] + 1
        d[:, 3] = d[:, 3] - d[:, 1] + 1
        rows = torch.cat([labels[:, None].to(d), d[:, [4, 0, 1, 2, 3]]], dim=1).tolist()
        return [[int(r[0])] + r[1:] for r in rows]

    def _detect_with_retry(self, cls, reg, meta, score_thr, iou_thr, agn):
        out = self.detect((cls, reg), meta, score_thr, iou_thr, agn)
        counts = out.counts.cpu()
        if bool((counts[:, 2] != 0).any()):   # candidate capacity exceeded: rerun with the exact need
            need = int(counts[:, 3].max())
            out = self.detect((cls, reg), meta, score_thr, iou_thr, agn, max_candidates=need)
            counts = out.counts.cpu()
        return out, counts

    def get_results(self, predict_outputs, *args):
        """lfd.py:397-432: list (per image) of [label, score, x1, y1, w, h] rows."""
        cls, reg = predict_outputs
        meta_batch = args[0]
        meta = torch.tensor([[float(m['resized_width']), float(m['resized_height']), float(m['resize_scale'])]
                             for m in meta_batch], dtype=torch.float32, device=cls.device)
        out, counts = self._detect_with_retry(cls, reg, meta, self._classification_threshold,
                                              self._nms_cfg.get('iou_thr', 0.5),
                                              self._nms_cfg.get('class_agnostic', False))
        results = []
        for i in range(cls.size(0)):
            k = int(counts[i, 1])
            results.append(self._pack(out.dets[i, :k], out.labels[i, :k]))
        return results

    def predict_for_single_image(self, image, aug_pipeline, classification_threshold=None, nms_threshold=None,
                                 class_agnostic=False, cuda_device_index=0):
        """lfd.py:544-655.  `image`: HWC numpy array (string paths need cv2, which the reference
        imports and this package does not)."""
        if isinstance(image, str):
            raise RuntimeError('predict_for_single_image: pass a decoded HWC numpy image (cv2 is not a dependency)')
        assert isinstance(image, numpy.ndarray)
        sample = {'image': image}
        sample = aug_pipeline(sample)
        data = numpy.ascontiguousarray(sample['image'][None].transpose([0, 3, 1, 2]))
        x = torch.from_numpy(data).float()
        H, W = x.size(2), x.size(3)
        x = x.cuda(cuda_device_index)
        self.cuda(cuda_device_index)
        self.eval()
        with torch.no_grad():
            cls, reg = self.forward_resident(x)
        thr = classification_threshold if classification_threshold is not None else self._classification_threshold
        if nms_threshold:                       # sticky mutation, as in the reference (:630-633)
            self._nms_cfg.update({'iou_thr': nms_threshold})
        if class_agnostic:
            self._nms_cfg.update({'class_agnostic': class_agnostic})
        meta = torch.tensor([[float(W), float(H), 1.0]], dtype=torch.float32, device=cls.device)
        out, counts = self._detect_with_retry(cls, reg, meta, thr, self._nms_cfg.get('iou_thr', 0.5),
                                              self._nms_cfg.get('class_agnostic', False))
        k = int(counts[0, 1])
        return self._pack(out.dets[0, :k], out.labels[0, :k])
