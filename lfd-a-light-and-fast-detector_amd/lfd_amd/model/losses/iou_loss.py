"""IoULoss / GIoULoss / DIoULoss / CIoULoss -- mirror of lfd/model/losses/iou_loss.py (bbox_overlaps :11-102, iou_loss
:105-123, giou_loss :127-169, diou_loss :172-223, ciou_loss :226-283, the four loss modules :286-430).  The losses and
their gradients run in csrc/losses.hip / csrc/boxloss.hip on GPU tensors; `bbox_overlaps` (the non-aligned helper, not
on the LFD path) stays tensor algebra."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from .utils import weighted_loss

__all__ = ['IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss', 'iou_loss', 'giou_loss', 'diou_loss', 'ciou_loss', 'bbox_overlaps']


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


class _BoxLossFunction(Function):
    """loss [n] of lfd_box_loss_f32; the kernel returns the gradient w.r.t. the predicted box with the loss (forward-mode
    differentiation in the kernel), the backward is one multiply."""

    @staticmethod
    def forward(ctx, pred, target, kind, eps):
        loss, grad = ops.box_loss(pred, target, kind, eps, want_grad=pred.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        (grad,) = ctx.saved_tensors
        return grad * d_loss.unsqueeze(-1), None, None, None


@weighted_loss
def giou_loss(pred, target, eps=1e-7):
    """1 - GIoU (iou_loss.py:127-169)"""
    return _BoxLossFunction.apply(pred, target, 'giou', eps)


@weighted_loss
def diou_loss(pred, target, eps=1e-7):
    """1 - IoU + rho^2 / c^2 (iou_loss.py:172-223)"""
    return _BoxLossFunction.apply(pred, target, 'diou', eps)


@weighted_loss
def ciou_loss(pred, target, eps=1e-7):
    """DIoU + the aspect-ratio term v^2 / (1 - IoU + v) (iou_loss.py:226-283)"""
    return _BoxLossFunction.apply(pred, target, 'ciou', eps)


class _UnionLoss(nn.Module):
    """Shared module shell of GIoULoss / DIoULoss / CIoULoss (iou_loss.py:324-430): same constructor, all-zero weights
    return `(pred * weight).sum()` BEFORE the reduction_override check, [n,4] weights are averaged to [n]."""
    _fn = None

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            return (pred * weight).sum()
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * type(self)._fn(pred, target, weight, eps=self.eps, reduction=reduction,
                                                 avg_factor=avg_factor, **kwargs)


class GIoULoss(_UnionLoss):
    _fn = staticmethod(giou_loss)


class DIoULoss(_UnionLoss):
    _fn = staticmethod(diou_loss)


class CIoULoss(_UnionLoss):
    _fn = staticmethod(ciou_loss)
