"""oracle/ref_import.py -- TEST INFRASTRUCTURE (build container only).

Makes the reference's own Python modules importable in this container so that
tests/golden/make_golden.py can run the REAL reference on seeded inputs and freeze its
outputs as fixtures.  /root/reference does not exist on the GPU box; nothing under
tests/ (other than the generator script), bench.py or smoke() calls this at run time.

Why stubs are needed (reference file:line):
  * lfd/model/lfd.py:6,10 import cv2 and pycuda.driver at module level (absent here);
  * lfd/model/lfd.py:8 pulls lfd.data_pipeline (albumentations, pycocotools, turbojpeg
    crash at dataset/utils/turbojpeg.py:451-456) -> pre-seed a stub that exposes Sample
    loaded by path from data_pipeline/dataset/sample.py;
  * lfd/model/utils/nms.py:4 needs libs/nms_ext -> the reference's own CPU extension
    compiled unmodified (oracle/build_ref.py);
  * lfd/model/losses/focal_loss.py:6 needs sigmoid_focal_loss_ext, which has NO CPU path in
    the reference (sigmoid_focal_loss_ext.cpp:32,49) -> served by oracle/lfd_oracle.c's
    restatement of sigmoid_focal_loss_cuda.cu:24-97.
"""
import importlib.util
import sys
import types

REF_ROOT = '/root/reference'


def available():
    import os
    return os.path.isdir(REF_ROOT + '/lfd')


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference's `lfd.model` package (real code, CPU)."""
    if 'lfd.model' in sys.modules and getattr(sys.modules['lfd.model'], '_lfd_ref_marker', False):
        return sys.modules['lfd.model']
    import torch
    from . import build_ref, c_oracle
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for n in ('cv2', 'pycuda', 'pycuda.driver'):
        if n not in sys.modules:
            _stub(n)
    sp = importlib.util.spec_from_file_location(
        '_lfd_ref_sample', REF_ROOT + '/lfd/data_pipeline/dataset/sample.py')
    sm = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(sm)
    _stub('lfd.data_pipeline').__path__ = []
    _stub('lfd.data_pipeline.dataset', Sample=sm.Sample, reserved_keys=sm.reserved_keys)
    sys.modules['lfd.model.utils.libs.nms_ext'] = build_ref.load_ref()

    def _fwd(logits, targets, num_classes, gamma, alpha):
        assert logits.dim() == 2 and logits.size(1) == num_classes
        out = c_oracle.sigmoid_focal_loss_fwd(logits.detach().float().numpy(), targets.numpy(),
                                              gamma, alpha)
        return torch.from_numpy(out).to(logits.dtype)

    def _bwd(logits, targets, d_losses, num_classes, gamma, alpha):
        out = c_oracle.sigmoid_focal_loss_bwd(logits.detach().float().numpy(), targets.numpy(),
                                              d_losses.detach().float().numpy(), gamma, alpha)
        return torch.from_numpy(out).to(logits.dtype)

    _stub('lfd.model.losses.libs.sigmoid_focal_loss_ext', forward=_fwd, backward=_bwd)
    import lfd.model as M
    M._lfd_ref_marker = True
    return M
