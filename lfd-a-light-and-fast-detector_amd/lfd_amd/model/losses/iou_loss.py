"""IoULoss -- mirror of lfd/model/losses/iou_loss.py (bbox_overlaps :11-102, iou_loss :105-123,
IoULoss :286-321).  The aligned-IoU -log loss and its gradient run in csrc/losses.hip on GPU
tensors; `bbox_overlaps` (the non-aligned helper, not on the LFD path) stays tensor algebra."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from .utils import weighted_loss

__all__ = ['IoULoss', 'iou_loss', 'bbox_overlaps']


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    assert mode in ['iou', 'iof']
    assert bboxes1.size(-1) == 4 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 4 or bboxes2.size(0) == 0
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(rows, 1) if is_aligned else bboxes1.new(rows, cols)
    a1 = (bboxes1[:, 2] - bboxes1[:, 0]) * (bboxes1[:, 3] - bboxes1[:, 1])
    a2 = (bboxes2[:, 2] - bboxes2[:, 0]) * (bboxes2[:, 3] - bboxes2[:, 1])
    if is_aligned:
        lt = torch.max(bboxes1[:, :2], bboxes2[:, :2])
        rb = torch.min(bboxes1[:, 2:], bboxes2[:, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        union = a1 + a2 - overlap if mode == 'iou' else a1
    else:
        lt = torch.max(bboxes1[:, None, :2], bboxes2[:, :2])
        rb = torch.min(bboxes1[:, None, 2:], bboxes2[:, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[:, :, 0] * wh[:, :, 1]
        union = a1[:, None] + a2 - overlap if mode == 'iou' else a1[:, None]
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


class _IoULossFunction(Function):
    @staticmethod
    def forward(ctx, pred, target, eps):
        ctx.save_for_backward(pred, target)
        ctx.eps = eps
        return ops.iou_loss_forward(pred, target, eps)

    @staticmethod
    def backward(ctx, d_loss):
        pred, target = ctx.saved_tensors
        return ops.iou_loss_backward(pred, target, d_loss.contiguous(), ctx.eps), None, None


@weighted_loss
def iou_loss(pred, target, eps=1e-6):
    """loss = -log(clamp(aligned IoU, eps))  (iou_loss.py:121-123)"""
    return _IoULossFunction.apply(pred, target, eps)


class IoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if (weight is not None) and (not torch.any(weight > 0)) and (reduction != 'none'):
            return (pred * weight).sum()
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * iou_loss(pred, target, weight, eps=self.eps, reduction=reduction,
                                           avg_factor=avg_factor, **kwargs)
