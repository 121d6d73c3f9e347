"""FocalLoss -- mirror of lfd/model/losses/focal_loss.py:12-92 over the HIP focal kernels."""
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .libs import sigmoid_focal_loss_ext
from .utils import weight_reduce_loss

__all__ = ['FocalLoss', 'sigmoid_focal_loss']


class SigmoidFocalLossFunction(Function):
    @staticmethod
    def forward(ctx, input, target, gamma=2.0, alpha=0.25):
        ctx.save_for_backward(input, target)
        ctx.num_classes, ctx.gamma, ctx.alpha = input.shape[1], gamma, alpha
        return sigmoid_focal_loss_ext.forward(input, target, input.shape[1], gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        input, target = ctx.saved_tensors
        d_input = sigmoid_focal_loss_ext.backward(input, target, d_loss.contiguous(), ctx.num_classes, ctx.gamma,
                                                  ctx.alpha)
        return d_input, None, None, None, None


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    loss = SigmoidFocalLossFunction.apply(pred, target, gamma, alpha)
    if weight is not None:
        weight = weight.view(-1, 1)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)
