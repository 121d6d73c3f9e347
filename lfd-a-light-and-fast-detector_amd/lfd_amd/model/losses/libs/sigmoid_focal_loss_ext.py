"""Drop-in for `lfd.model.losses.libs.sigmoid_focal_loss_ext`
(lfd/model/losses/build/sigmoid_focal_loss/src/sigmoid_focal_loss_ext.cpp:52-57), ctypes over liblfd_hip.so ONLY -- no
import from lfd_amd, so the file can be copied to lfd/model/losses/libs/ of the reference tree as it is (INTEGRATION.md
section 2).  The library is looked up in $LFD_HIP_LIB, next to this file, then in the lfd_amd package directory.

  forward(logits[N,C] f32|f16, targets[N] int64, num_classes, gamma, alpha) -> losses[N,C]
  backward(logits, targets, d_losses[N,C], num_classes, gamma, alpha) -> d_logits[N,C]
GPU-only, like the reference (:32,49 raise on CPU tensors); kernels: csrc/losses.hip (restating
sigmoid_focal_loss_cuda.cu:24-97)."""
import ctypes as C
import os

import torch

_lib = None
_ABI_VERSION = 3      # LFD_HIP_ABI_VERSION of include/lfd_hip.h


def _load():
    global _lib
    if _lib is None:
        here = os.path.dirname(os.path.abspath(__file__))
        cands = [os.environ.get('LFD_HIP_LIB'), os.path.join(here, 'liblfd_hip.so'),
                 os.path.join(here, '..', '..', '..', 'liblfd_hip.so')]
        path = next((p for p in cands if p and os.path.exists(p)), None)
        if path is None:
            raise RuntimeError('sigmoid_focal_loss_ext: liblfd_hip.so not found (set LFD_HIP_LIB or build it with '
                               '`python __graft_entry__.py`)')
        l = C.CDLL(path)
        l.lfd_hip_abi_version.restype = C.c_int
        if l.lfd_hip_abi_version() != _ABI_VERSION:
            raise RuntimeError('sigmoid_focal_loss_ext: %s has ABI version %d, this file binds version %d'
                               % (path, l.lfd_hip_abi_version(), _ABI_VERSION))
        l.lfd_sigmoid_focal_loss_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                                 C.c_int32, C.c_void_p]
        l.lfd_sigmoid_focal_loss_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                                 C.c_void_p, C.c_int32, C.c_void_p]
        _lib = l
    return _lib


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise RuntimeError('sigmoid_focal_loss: float32 / float16 logits only (got %s)' % t.dtype)


def _targets(targets, logits):
    t = targets.to(device=logits.device, dtype=torch.long).contiguous()
    if t.dim() != 1 or t.size(0) != logits.size(0):
        raise RuntimeError('targets should be N')
    return t


def forward(logits, targets, num_classes, gamma, alpha):
    if not logits.is_cuda:
        raise RuntimeError('SigmoidFocalLoss is not implemented on the CPU')      # sigmoid_focal_loss_ext.cpp:32
    if logits.dim() != 2:
        raise RuntimeError('logits should be NxClass')                             # .cu:105
    if logits.size(1) != num_classes:
        raise RuntimeError('logits.size(1) must equal num_classes')
    x = logits.detach().contiguous()                                               # .cu:124-125
    t = _targets(targets, x)
    out = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            rc = _load().lfd_sigmoid_focal_loss_fwd(x.data_ptr(), t.data_ptr(), x.size(0), x.size(1), float(gamma), float(alpha),
                                                    out.data_ptr(), _dtype_code(x),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError('lfd_sigmoid_focal_loss_fwd failed (status %d)' % rc)
    return out


def backward(logits, targets, d_losses, num_classes, gamma, alpha):
    if not logits.is_cuda:
        raise RuntimeError('SigmoidFocalLoss is not implemented on the CPU')      # sigmoid_focal_loss_ext.cpp:49
    if logits.dim() != 2 or logits.size(1) != num_classes:
        raise RuntimeError('logits should be NxClass')                             # .cu:145-146
    x = logits.detach().contiguous()
    t = _targets(targets, x)
    d = d_losses.detach().to(x.dtype).contiguous()
    if d.shape != x.shape:
        raise RuntimeError('d_losses must have the shape of logits')
    out = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            rc = _load().lfd_sigmoid_focal_loss_bwd(x.data_ptr(), t.data_ptr(), d.data_ptr(), x.size(0), x.size(1), float(gamma),
                                                    float(alpha), out.data_ptr(), _dtype_code(x),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError('lfd_sigmoid_focal_loss_bwd failed (status %d)' % rc)
    return out
