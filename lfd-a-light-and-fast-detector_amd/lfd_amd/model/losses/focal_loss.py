"""FocalLoss -- host-side mirror of the reference's operator interface (lfd/model/losses/focal_loss.py:12-92):
`FocalLoss(use_sigmoid, gamma, alpha, reduction, loss_weight)`, `sigmoid_focal_loss(pred, target, weight, ...)` and the
autograd function over the extension module's forward / backward -- which here are the HIP kernels of csrc/losses.hip
behind liblfd_hip.so (no CPU implementation, like the reference's CUDA-only extension).
"""
import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from .libs import sigmoid_focal_loss_ext as _ext
from .utils import weight_reduce_loss

__all__ = ['FocalLoss', 'sigmoid_focal_loss']


class SigmoidFocalLossFunction(torch.autograd.Function):
    """elementwise loss [N, C] of logits [N, C] against integer labels [N] (label C = background)"""

    @staticmethod
    def forward(ctx, logits, labels, gamma=2.0, alpha=0.25):
        channels = logits.shape[1]
        ctx.hyper = (channels, gamma, alpha)
        ctx.save_for_backward(logits, labels)
        return _ext.forward(logits, labels, channels, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_losses):
        channels, gamma, alpha = ctx.hyper
        logits, labels = ctx.saved_tensors
        grad_logits = _ext.backward(logits, labels, grad_losses.contiguous(), channels, gamma, alpha)
        return grad_logits, None, None, None, None


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    per_element = SigmoidFocalLossFunction.apply(pred, target, gamma, alpha)
    per_row_weight = None if weight is None else weight.view(-1, 1)
    return weight_reduce_loss(per_element, per_row_weight, reduction, avg_factor)


class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid = use_sigmoid
        self.gamma, self.alpha = gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        how = self.reduction if not reduction_override else reduction_override
        loss = sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha, reduction=how,
                                  avg_factor=avg_factor)
        return self.loss_weight * loss
