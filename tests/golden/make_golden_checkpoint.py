"""tests/golden/make_golden_checkpoint.py -- a checkpoint file written by the REFERENCE's own save_checkpoint
(lfd/execution/utils.py:90-122) from the reference's own modules, as a fixture (build container only).

    python tests/golden/make_golden_checkpoint.py

Writes tests/golden/ref_checkpoint_tiny.pth (~0.4 MB: a one-stage LFD so that the fixture stays small; same classes, same
key naming incl. the head's per-level keys) and ref_checkpoint_tiny.npz (the reference's eval-mode outputs for a seeded
input with those weights).  The model is wrapped in nn.DataParallel before saving, like Executor does (executor.py:39), the
optimizer / lr scheduler states ride along like CheckpointHook -> Executor.save (executor.py:126-132).
tests/test_host_logic.py loads the file with lfd_amd.checkpoint.load_checkpoint(strict=True); tests/test_gpu_forward.py
runs the loaded weights through the HIP engine against the stored outputs.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
warnings.filterwarnings('ignore')

from oracle import ref_import  # noqa: E402
from lfd_amd import configs  # noqa: E402

TINY = dict(configs.ARCHS['WIDERFACE_LFD_XS'], body_architecture=[1], body_channels=[64], out_indices=((0, 0),),
            regression_ranges=((4, 320),))


def main():
    M = ref_import.import_reference()
    import lfd.model.backbone as RB
    import lfd.model.head as RH
    import lfd.model.losses as RL
    import lfd.model.neck as RN
    for n in ('cv2', 'torchvision'):                   # lfd/execution/utils.py:11,13 import them at module level
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    import importlib.util
    sp = importlib.util.spec_from_file_location('_ref_exec_utils', '/root/reference/lfd/execution/utils.py')
    ru = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(ru)
    model = configs.build_modules(TINY, RB.LFDResNet, RN.SimpleNeck, RH.LFDHead, M.LFD, RL.FocalLoss, RL.IoULoss,
                                  RL.CrossEntropyLoss, seed=666)
    configs.perturb_weights(model, seed=2)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[60, 90], gamma=0.1)
    dp = torch.nn.DataParallel(model)
    path = os.path.join(HERE, 'ref_checkpoint_tiny.pth')
    ru.save_checkpoint(dp, path, optimizer=opt, lr_scheduler=sched, meta=dict(epoch=7))
    model.eval()
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(9)) * 2 - 1
    with torch.no_grad():
        cls, reg = model(x)
    np.savez_compressed(os.path.join(HERE, 'ref_checkpoint_tiny.npz'), cls=cls.numpy(), reg=reg.numpy(), x_seed=9,
                        shape=np.array([1, 64, 96]))
    print(path, os.path.getsize(path), 'bytes; outputs', tuple(cls.shape), tuple(reg.shape))


if __name__ == '__main__':
    main()
