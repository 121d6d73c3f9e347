"""Drop-in for the reference's pybind module `lfd.model.utils.libs.nms_ext`
(lfd/model/utils/build/nms/src/nms_ext.cpp:45-49): same three functions, same argument
meaning and error behaviour (RuntimeError), implemented over liblfd_hip.so's C ABI.

  nms(dets[n,5] f32 {x1,y1,x2,y2,score}, thr) -> LongTensor[k]   kept indices, score-descending
  soft_nms / nms_match: CPU-only in the reference (nms_ext.cpp:29-43 raise on GPU tensors) and
  not on any shipped config's path; kept as "not implemented on GPU" errors here.
"""
import torch

from ....ops import nms_indices


def nms(dets, threshold):
    if not isinstance(dets, torch.Tensor):
        raise TypeError('dets must be a Tensor')
    if not dets.is_cuda:
        raise RuntimeError('nms: this build provides the MI355X (HIP) implementation only; got a CPU tensor')
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.long)   # nms_cuda.cpp:10-11 returns an empty CPU long
    return nms_indices(dets, threshold)


def soft_nms(dets, threshold, method, sigma, min_score):
    raise RuntimeError('soft_nms is not implemented on GPU')   # nms_ext.cpp:33


def nms_match(dets, threshold):
    raise RuntimeError('nms_match is not implemented on GPU')  # nms_ext.cpp:40
