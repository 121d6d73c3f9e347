"""NMS API -- mirror of lfd/model/utils/nms.py (nms :7-59, batched_nms :119-158,
multiclass_nms :161-220) over liblfd_hip.so.  batched_nms / multiclass_nms are the hot path and
need tensors on the MI355X (sort, IoU bitmask, greedy scan, class-offset trick incl. its fp32
rounding: csrc/postproc.hip); plain nms also serves CPU tensors / numpy arrays and soft_nms is
CPU-only, exactly like the reference's extension (host routines: csrc/nms_host.hip)."""
import numpy as np
import torch

from ... import ops
from .libs import nms_ext

__all__ = ['nms', 'soft_nms', 'batched_nms', 'multiclass_nms']


def nms(dets, iou_thr, device_id=None):
    """Returns (dets[inds], inds) like the reference (:7-59): tensors are processed on the device they live on -- GPU
    tensors by the HIP kernels, CPU tensors by the host routine of the same library (nms_ext.cpp:18-27 dispatches the same
    way); numpy arrays on the CPU unless `device_id` is given (:36-47)."""
    if isinstance(dets, torch.Tensor):
        is_numpy, dets_th = False, dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        device = 'cpu' if device_id is None else 'cuda:{}'.format(device_id)
        dets_th = torch.from_numpy(dets).to(device)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(type(dets)))
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        inds = nms_ext.nms(dets_th, iou_thr)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """CPU-only in the reference (:62-116): tensors are moved to the host, results returned in the input's type / device."""
    if isinstance(dets, torch.Tensor):
        is_tensor, dets_t = True, dets.detach().cpu()
    elif isinstance(dets, np.ndarray):
        is_tensor, dets_t = False, torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(type(dets)))
    method_codes = {'linear': 1, 'gaussian': 2}
    if method not in method_codes:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))
    results = nms_ext.soft_nms(dets_t, iou_thr, method_codes[method], sigma, min_score)
    new_dets, inds = results[:, :5], results[:, 5]
    if is_tensor:
        return new_dets.to(device=dets.device, dtype=dets.dtype), inds.to(device=dets.device, dtype=torch.long)
    return new_dets.numpy().astype(dets.dtype), inds.numpy().astype(np.int64)


def batched_nms(bboxes, scores, inds, nms_cfg, class_agnostic=False):
    """One fused device pass instead of offsets + cat + nms + subtract (:143-156)."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    nms_type = cfg.pop('type', 'nms')
    if nms_type != 'nms':
        raise RuntimeError('batched_nms: only type="nms" is implemented on the MI355X')
    dets, keep = ops.batched_nms_dets(bboxes, scores, inds, cfg.get('iou_thr', 0.5), class_agnostic)
    return dets, keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """Same contract as the reference (:161-220): multi_scores' last column is background and
    ignored; returns (dets[k,5], labels[k]); labels live on the CPU like the reference's
    (:196-197) when candidates exist."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 4), scores.reshape(-1), labels.reshape(-1)
    inds = (scores > score_thr).nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if inds.numel() == 0:
        return bboxes, labels.cpu()
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep].cpu()
