"""Drop-in for `lfd.model.losses.libs.sigmoid_focal_loss_ext`
(lfd/model/losses/build/sigmoid_focal_loss/src/sigmoid_focal_loss_ext.cpp:52-57):
forward(logits[N,C], targets[N] int64, num_classes, gamma, alpha) -> losses[N,C]
backward(logits, targets, d_losses, num_classes, gamma, alpha) -> d_logits[N,C]
GPU-only, like the reference (:32,49 raise on CPU tensors)."""
from ....ops import focal_backward, focal_forward


def forward(logits, targets, num_classes, gamma, alpha):
    if not logits.is_cuda:
        raise RuntimeError('SigmoidFocalLoss is not implemented on the CPU')
    if logits.dim() != 2:
        raise RuntimeError('logits should be NxClass')
    if logits.size(1) != num_classes:
        raise RuntimeError('logits.size(1) must equal num_classes')
    return focal_forward(logits, targets, gamma, alpha)


def backward(logits, targets, d_losses, num_classes, gamma, alpha):
    if not logits.is_cuda:
        raise RuntimeError('SigmoidFocalLoss is not implemented on the CPU')
    if logits.dim() != 2 or logits.size(1) != num_classes:
        raise RuntimeError('logits should be NxClass')
    return focal_backward(logits, targets, d_losses, gamma, alpha)
