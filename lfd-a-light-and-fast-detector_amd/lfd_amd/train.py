"""One training iteration -- the loop body of Executor.train (lfd/execution/executor.py:191-211) with
OptimizerHook.after_train_iter (lfd/execution/hooks/optimizer_hook.py:26-36), image-parallel over ranks instead of
nn.DataParallel (executor.py:39): forward -> get_loss (global-batch normalisers) -> zero_grad -> backward ->
one all-reduce of the flat gradient buffer -> clip_grad_norm_ (first `duration` epochs) + SGD update.
"""
from . import optim, parallel


def train_step(model, optimizer, image_batch, annotation_batch, grad_clip_cfg=None, clip_active=True):
    """-> (loss_values dict of floats, grad_norm).  grad_clip_cfg: dict(max_norm=..., norm_type=2) or None
    (config_dict['optimizer_grad_clip_cfg'] without 'duration'); clip_active: epoch < duration."""
    predict_outputs = model(image_batch)
    loss_dict = model.get_loss(predict_outputs, annotation_batch)
    grad_norm = backward_and_update(optimizer, loss_dict['loss'], grad_clip_cfg, clip_active)
    return loss_dict['loss_values'], grad_norm


def backward_and_update(optimizer, loss, grad_clip_cfg=None, clip_active=True):
    optimizer.zero_grad()
    loss.backward()
    fused = isinstance(optimizer, optim.SGD)
    if fused:
        optimizer.allreduce_grads()
    elif parallel.is_dist():      # any other optimizer: average the gradients over the ranks as one flat bucket
        parallel.allreduce_mean_([p.grad for g in optimizer.param_groups for p in g['params'] if p.grad is not None])
    grad_norm = 0
    if grad_clip_cfg is not None and clip_active:
        if fused and float(grad_clip_cfg.get('norm_type', 2)) == 2.0:
            return optimizer.clip_and_step(grad_clip_cfg['max_norm'])
        import torch.nn.utils.clip_grad as clip_grad
        params = [p for g in optimizer.param_groups for p in g['params'] if p.requires_grad and p.grad is not None]
        if params:
            grad_norm = clip_grad.clip_grad_norm_(params, **grad_clip_cfg)
    optimizer.step()
    return grad_norm


class OptimizerHook(object):
    """Same constructor and after_train_iter(executor) contract as the reference hook (optimizer_hook.py:8-36):
    reads executor.config_dict['optimizer' | 'loss' | 'model' | 'epoch'], writes config_dict['grad_norm']."""

    def __init__(self, grad_clip_cfg, training_epochs):
        assert isinstance(grad_clip_cfg, dict) or grad_clip_cfg is None
        self._grad_clip_cfg = dict(grad_clip_cfg) if grad_clip_cfg is not None else None
        if self._grad_clip_cfg is not None:
            self._grad_clip_duration = self._grad_clip_cfg.pop('duration', training_epochs)
            assert self._grad_clip_duration > 0 and isinstance(self._grad_clip_duration, int)

    def after_train_iter(self, executor):
        cd = executor.config_dict
        active = self._grad_clip_cfg is not None and cd['epoch'] < self._grad_clip_duration
        cd['grad_norm'] = backward_and_update(cd['optimizer'], cd['loss'], self._grad_clip_cfg, active)
