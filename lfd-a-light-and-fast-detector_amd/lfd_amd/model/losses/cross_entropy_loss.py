"""CrossEntropyLoss -- mirror of lfd/model/losses/cross_entropy_loss.py:12-50 (TT100K configs)."""
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from .utils import weight_reduce_loss

__all__ = ['CrossEntropyLoss']


class _CrossEntropyFunction(Function):
    """F.cross_entropy(pred, label, reduction='none') through liblfd_hip.so (lfd_cross_entropy_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, pred, label):
        ctx.save_for_backward(pred, label)
        return ops.cross_entropy_forward(pred, label)

    @staticmethod
    def backward(ctx, d_loss):
        pred, label = ctx.saved_tensors
        return ops.cross_entropy_backward(pred, label, d_loss.contiguous()), None


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None):
    loss = _CrossEntropyFunction.apply(pred, label)
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


class CrossEntropyLoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * cross_entropy(cls_score, label, weight, reduction=reduction, avg_factor=avg_factor)
