"""SmoothL1Loss / L1Loss / MSELoss -- host-side mirror of lfd/model/losses/smooth_l1_loss.py:11-130 and mse_loss.py:11-50,
LFD's "independent" regression-loss family (lfd.py:61-66).  Loss and derivative come from one launch of
lfd_pointwise_loss_f32 (csrc/boxloss.hip); weighting / reduction is the shared `weighted_loss` wrapper.
"""
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from .utils import weighted_loss

__all__ = ['SmoothL1Loss', 'L1Loss', 'MSELoss', 'smooth_l1_loss', 'l1_loss', 'mse_loss']


class _PointwiseLossFunction(Function):
    @staticmethod
    def forward(ctx, pred, target, kind, beta):
        loss, grad = ops.pointwise_loss(pred, target, kind, beta, want_grad=pred.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        (grad,) = ctx.saved_tensors
        return grad * d_loss, None, None, None


def _checked(pred, target):
    assert pred.size() == target.size() and target.numel() > 0


@weighted_loss
def smooth_l1_loss(pred, target, beta=1.0):
    assert beta > 0
    _checked(pred, target)
    return _PointwiseLossFunction.apply(pred, target, 'smooth_l1', beta)


@weighted_loss
def l1_loss(pred, target):
    _checked(pred, target)
    return _PointwiseLossFunction.apply(pred, target, 'l1', 1.0)


@weighted_loss
def mse_loss(pred, target):
    return _PointwiseLossFunction.apply(pred, target, 'mse', 1.0)


def _reduction(module, override):
    assert override in (None, 'none', 'mean', 'sum')
    return override if override else module.reduction


class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta,
                                                 reduction=_reduction(self, reduction_override), avg_factor=avg_factor,
                                                 **kwargs)


class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        return self.loss_weight * l1_loss(pred, target, weight, reduction=_reduction(self, reduction_override),
                                          avg_factor=avg_factor)


class MSELoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        return self.loss_weight * mse_loss(pred, target, weight, reduction=self.reduction, avg_factor=avg_factor)
