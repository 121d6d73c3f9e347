"""weight_reduce_loss / weighted_loss -- mirror of lfd/model/losses/utils.py:9-100."""
import functools

import torch.nn.functional as F


def reduce_loss(loss, reduction):
    code = F._Reduction.get_enum(reduction)   # none 0, mean 1, sum 2
    if code == 0:
        return loss
    return loss.mean() if code == 1 else loss.sum()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def weighted_loss(loss_func):
    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        return weight_reduce_loss(loss_func(pred, target, **kwargs), weight, reduction, avg_factor)
    return wrapper
