"""Checkpoint files in the reference's format -- mirror of load_checkpoint / save_checkpoint
(lfd/execution/utils.py:19-71, :90-122), so that `.pth` files written by the reference's CheckpointHook
(`{'meta', 'state_dict', 'optimizer_state_dict', 'lr_scheduler_state_dict'}`) load into this package's modules
with strict=True and files written here resume in the reference.

Behaviours kept: the 'module.' prefix of nn.DataParallel checkpoints is stripped (:47-49); a model wrapped in a
`.module` container is refused on load (:51) and unwrapped on save (:112); weights are saved on the CPU (:74-87);
`meta['time']` is stamped (:107); missing / unexpected keys are reported on rank 0 only (:54-69).
The state_dict itself needs no translation: parameter names, shapes (OIHW fp32) and the duplicated keys of the shared
head (`_head.head{i}_*` aliases of one tensor) are those of the reference modules; the fp16 NHWC inference plan is
rebuilt from the loaded parameters on the next forward (engine.get_plan watches the tensor versions).
"""
import os
import time
from collections import OrderedDict

import torch

from . import parallel

__all__ = ['load_checkpoint', 'save_checkpoint']


def load_checkpoint(model, load_path, map_location='cpu', strict=False, logger=None):
    if not os.path.isfile(load_path):
        raise IOError('{} is not a checkpoint file'.format(load_path))
    checkpoint = torch.load(load_path, map_location=map_location)
    if not (isinstance(checkpoint, dict) and 'state_dict' in checkpoint):
        raise RuntimeError('No state_dict found in checkpoint file {}'.format(load_path))
    state_dict = checkpoint['state_dict']
    if state_dict and next(iter(state_dict)).startswith('module.'):
        state_dict = OrderedDict((k[len('module.'):], v) for k, v in state_dict.items())
    assert not hasattr(model, 'module'), 'do not use DataParallel to wrap the model before loading state dict!'
    missing, unexpected = model.load_state_dict(state_dict, strict=strict)
    rank = torch.distributed.get_rank() if parallel.is_dist() else 0
    if rank == 0:
        say = logger.info if logger is not None else print
        if missing:
            say('[state dict loading warning] missing keys: {}'.format(','.join(missing)))
        if unexpected:
            say('[state dict loading warning] unexpected keys: {}'.format(','.join(unexpected)))
    return checkpoint


def save_checkpoint(model, save_path, optimizer=None, lr_scheduler=None, meta=None):
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError('meta must be a dict or None, but got {}'.format(type(meta)))
    meta.update(time=time.asctime())
    folder = os.path.dirname(save_path)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)
    source = model.module if hasattr(model, 'module') else model
    checkpoint = {'meta': meta,
                  'state_dict': OrderedDict((k, v.cpu()) for k, v in source.state_dict().items())}
    if optimizer is not None:
        checkpoint['optimizer_state_dict'] = optimizer.state_dict()
    if lr_scheduler is not None:
        checkpoint['lr_scheduler_state_dict'] = lr_scheduler.state_dict()
    torch.save(checkpoint, save_path)
