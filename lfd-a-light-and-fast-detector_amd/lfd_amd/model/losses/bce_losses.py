"""BCEWithLogitsLoss / QualityFocalLoss -- host-side mirror of lfd/model/losses/bce_with_logits_loss.py:12-80 and
gfocal_loss.py:11-141 (the two other classification losses LFD's constructor accepts, lfd.py:52-56).  Loss and derivative
come from one launch each (lfd_bce_with_logits_f32 / lfd_quality_focal_loss_f32, csrc/boxloss.hip).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from .utils import weight_reduce_loss, weighted_loss

__all__ = ['BCEWithLogitsLoss', 'QualityFocalLoss', 'binary_cross_entropy', 'quality_focal_loss']


class _BCEFunction(Function):
    @staticmethod
    def forward(ctx, logits, targets):
        loss, grad = ops.bce_with_logits(logits, targets, want_grad=logits.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        (grad,) = ctx.saved_tensors
        return grad * d_loss, None


class _QFLFunction(Function):
    @staticmethod
    def forward(ctx, logits, labels, scores, beta):
        loss, grad = ops.quality_focal_loss(logits, labels, scores, beta, want_grad=logits.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        (grad,) = ctx.saved_tensors
        return grad * d_loss.unsqueeze(-1), None, None, None


def _one_hot_from_1_based(labels, label_weights, channels):
    """integer labels (0 = background, k >= 1 -> channel k - 1) to a [N, channels] 0/1 target (bce_with_logits_loss.py:12-25)"""
    target = labels.new_zeros((labels.size(0), channels))
    fg = torch.nonzero(labels >= 1, as_tuple=False).reshape(-1)
    if fg.numel() > 0:
        target[fg, labels[fg] - 1] = 1
    w = None if label_weights is None else label_weights.view(-1, 1).expand(label_weights.size(0), channels)
    return target, w


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None):
    if pred.dim() != label.dim():
        label, weight = _one_hot_from_1_based(label, weight, pred.size(-1))
    loss = _BCEFunction.apply(pred, label.float())
    if weight is not None:
        loss = loss * weight.float()          # F.binary_cross_entropy_with_logits(..., weight): elementwise rescale
    return weight_reduce_loss(loss, reduction=reduction, avg_factor=avg_factor)


@weighted_loss
def quality_focal_loss(pred, target, beta=2.0):
    assert len(target) == 2, 'target for QFL must be a tuple of two elements: category label and quality label'
    label, score = target
    return _QFLFunction.apply(pred, label, score, beta)


class BCEWithLogitsLoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        how = reduction_override if reduction_override else self.reduction
        return self.loss_weight * binary_cross_entropy(cls_score, label, weight, reduction=how, avg_factor=avg_factor)


class QualityFocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, beta=2.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid in QFL supported now.'
        self.use_sigmoid, self.beta, self.reduction, self.loss_weight = use_sigmoid, beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        how = reduction_override if reduction_override else self.reduction
        return self.loss_weight * quality_focal_loss(pred, target, weight, beta=self.beta, reduction=how,
                                                     avg_factor=avg_factor)
