"""Drop-in for the reference's pybind module `lfd.model.utils.libs.nms_ext`
(lfd/model/utils/build/nms/src/nms_ext.cpp:45-49): same three functions, same argument meaning and error behaviour
(RuntimeError), implemented over liblfd_hip.so's C ABI with ctypes ONLY -- this file imports nothing from lfd_amd, so it
can be copied to lfd/model/utils/libs/nms_ext.py of the reference tree as it is.  The library
is looked up in $LFD_HIP_LIB, next to this file, then in the lfd_amd package directory.

  nms(dets[n,5] f32 {x1,y1,x2,y2,score}, thr) -> LongTensor[k]   kept indices, score-descending, on dets' device
        GPU tensor: lfd_nms_f32 (csrc/postproc.hip: device sort + 64x64 IoU bitmask + device greedy scan)
        CPU tensor: lfd_nms_cpu_f32 (csrc/nms_host.hip), like nms_ext.cpp:18-27 dispatches to cpu/nms_cpu.cpp:7-66
  soft_nms(dets, thr, method, sigma, min_score) -> Tensor[k,6]   CPU tensors only (nms_ext.cpp:29-36)
  nms_match(dets, thr) -> list[list[int]]                        CPU tensors only (nms_ext.cpp:38-43)
"""
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
            raise RuntimeError('nms_ext: liblfd_hip.so not found (set LFD_HIP_LIB or build it with `python __graft_entry__.py`)')
        l = C.CDLL(path)
        l.lfd_hip_abi_version.restype = C.c_int
        if l.lfd_hip_abi_version() != _ABI_VERSION:
            raise RuntimeError('nms_ext: %s has ABI version %d, this file binds version %d' % (path, l.lfd_hip_abi_version(), _ABI_VERSION))
        l.lfd_nms_workspace_bytes.restype = C.c_size_t
        l.lfd_nms_workspace_bytes.argtypes = [C.c_int64]
        l.lfd_nms_f32.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.lfd_nms_cpu_f32.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]
        l.lfd_soft_nms_cpu_f32.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        l.lfd_nms_match_cpu_f32.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        l.lfd_nms_cpu_f64.argtypes = l.lfd_nms_cpu_f32.argtypes
        l.lfd_soft_nms_cpu_f64.argtypes = l.lfd_soft_nms_cpu_f32.argtypes
        l.lfd_nms_match_cpu_f64.argtypes = l.lfd_nms_match_cpu_f32.argtypes
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (status %d)' % (what, rc))


def _host_dets(dets, what):
    """-> (contiguous CPU tensor, 'f32' | 'f64'): the reference evaluates a tensor in its own floating dtype
    (AT_DISPATCH_FLOATING_TYPES, nms_cpu.cpp:70,212,287); float64 -- numpy's default -- stays float64"""
    if dets.dim() != 2 or dets.size(1) != 5:
        raise RuntimeError('%s: dets must be [n,5]' % what)
    if dets.dtype == torch.float64:
        return dets.detach().contiguous(), 'f64'
    return dets.detach().contiguous().float(), 'f32'


def nms(dets, threshold):
    if not isinstance(dets, torch.Tensor):
        raise TypeError('dets must be a Tensor')
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.long)          # nms_cuda.cpp:10-11 / nms_cpu.cpp:11-13: empty CPU long
    if dets.dim() != 2 or dets.size(1) != 5:
        raise RuntimeError('nms: dets must be [n,5]')
    l = _load()
    n = dets.size(0)
    if not dets.is_cuda:
        d, sfx = _host_dets(dets, 'nms')
        keep = torch.empty(n, dtype=torch.long)
        num = C.c_int64(0)
        _check(getattr(l, 'lfd_nms_cpu_' + sfx)(d.data_ptr(), n, float(threshold), keep.data_ptr(), C.byref(num)), 'lfd_nms_cpu_' + sfx)
        return keep[:num.value]
    d = dets.detach().contiguous().float()
    with torch.cuda.device(d.device):
        keep = torch.empty(n, dtype=torch.long, device=d.device)
        num = torch.zeros(1, dtype=torch.int32, device=d.device)
        ws = torch.empty(max(int(l.lfd_nms_workspace_bytes(n)), 1), dtype=torch.uint8, device=d.device)
        _check(l.lfd_nms_f32(d.data_ptr(), n, float(threshold), keep.data_ptr(), num.data_ptr(), ws.data_ptr(), ws.numel(),
                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'lfd_nms_f32')
        k = int(num.item())       # data-dependent output length: the reference's GPU path synchronises too (nms_kernel.cu:105-111)
    return keep[:k]


def soft_nms(dets, threshold, method, sigma, min_score):
    if dets.is_cuda:
        raise RuntimeError('soft_nms is not implemented on GPU')       # nms_ext.cpp:33
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.long)                        # nms_cpu.cpp:83-85
    d, sfx = _host_dets(dets, 'soft_nms')
    n = d.size(0)
    out = torch.empty((n, 6), dtype=d.dtype)
    num = C.c_int64(0)
    _check(getattr(_load(), 'lfd_soft_nms_cpu_' + sfx)(d.data_ptr(), n, float(threshold), int(method), float(sigma), float(min_score),
                                                       out.data_ptr(), C.byref(num)), 'lfd_soft_nms_cpu_' + sfx)
    return out[:num.value].to(dets.dtype)


def nms_match(dets, threshold):
    if dets.is_cuda:
        raise RuntimeError('nms_match is not implemented on GPU')      # nms_ext.cpp:40
    if dets.numel() == 0:
        return []
    d, sfx = _host_dets(dets, 'nms_match')
    n = d.size(0)
    members = torch.empty(n, dtype=torch.int32)
    sizes = torch.empty(n, dtype=torch.int32)
    num = C.c_int64(0)
    _check(getattr(_load(), 'lfd_nms_match_cpu_' + sfx)(d.data_ptr(), n, float(threshold), members.data_ptr(), sizes.data_ptr(),
                                                        C.byref(num)), 'lfd_nms_match_cpu_' + sfx)
    out, o = [], 0
    mem = members.tolist()
    for s in sizes[:num.value].tolist():
        out.append(mem[o:o + s])
        o += s
    return out
