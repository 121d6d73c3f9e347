"""ctypes view of oracle/lfd_oracle.c (plain-C restatement).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liblfd_oracle.so')
_lib = None


def build():
    src = os.path.join(_HERE, 'lfd_oracle.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_nms_f32.restype = C.c_int64
        _lib.oracle_batched_nms_f32.restype = C.c_int64
        _lib.oracle_multiclass_nms_f32.restype = C.c_int64
        _lib.oracle_soft_nms_f32.restype = C.c_int64
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def argsort_desc_stable(scores):
    s = _f(scores)
    o = np.empty(s.shape[0], np.int64)
    lib().oracle_argsort_desc_stable(_p(s), C.c_int64(s.shape[0]), _p(o))
    return o


def nms(dets, thr):
    """nms_ext.nms semantics (nms_cpu.cpp:7-66): kept original indices, score-descending."""
    d = _f(dets).reshape(-1, 5)
    keep = np.empty(max(d.shape[0], 1), np.int64)
    n = lib().oracle_nms_f32(_p(d), C.c_int64(d.shape[0]), C.c_float(thr), _p(keep))
    return keep[:n].copy()


def batched_nms(boxes, scores, labels, iou_thr, class_agnostic=False):
    b = _f(boxes).reshape(-1, 4); s = _f(scores); l = np.ascontiguousarray(labels, np.int64)
    k = b.shape[0]
    dets = np.empty((max(k, 1), 5), np.float32); keep = np.empty(max(k, 1), np.int64)
    n = lib().oracle_batched_nms_f32(_p(b), _p(s), _p(l), C.c_int64(k), C.c_float(iou_thr),
                                     C.c_int(int(class_agnostic)), _p(dets), _p(keep))
    return dets[:n].copy(), keep[:n].copy()


def multiclass_nms(boxes, scores, score_thr, iou_thr, class_agnostic=False, max_num=-1):
    """multiclass_nms (nms.py:161-220) for boxes [n,4], scores [n,ncls] (bg column dropped).
    returns dets[k,5], labels[k], candidate ordinals[k], num_candidates."""
    b = _f(boxes).reshape(-1, 4); s = _f(scores)
    n, ncls = s.shape
    cap = max(n * ncls, 1)
    dets = np.empty((cap, 5), np.float32); labels = np.empty(cap, np.int64)
    cand = np.empty(cap, np.int64); nc = C.c_int64(0)
    k = lib().oracle_multiclass_nms_f32(_p(b), _p(s), C.c_int64(n), C.c_int64(ncls),
                                        C.c_float(score_thr), C.c_float(iou_thr),
                                        C.c_int(int(class_agnostic)), C.c_int64(max_num),
                                        _p(dets), _p(labels), _p(cand), C.byref(nc))
    if nc.value == 0:   # empty-candidate early return: (bboxes[0,4], labels[0]), not [0,5]  (nms.py:207-212)
        return np.zeros((0, 4), np.float32), labels[:0].copy(), cand[:0].copy(), 0
    return dets[:k].copy(), labels[:k].copy(), cand[:k].copy(), int(nc.value)


def soft_nms(dets, thr, method=1, sigma=0.5, min_score=1e-3):
    d = _f(dets).reshape(-1, 5)
    out = np.empty((max(d.shape[0], 1), 6), np.float32)
    n = lib().oracle_soft_nms_f32(_p(d), C.c_int64(d.shape[0]), C.c_float(thr), C.c_int(method),
                                  C.c_float(sigma), C.c_float(min_score), _p(out))
    return out[:n].copy()


def sigmoid_focal_loss_fwd(logits, targets, gamma=2.0, alpha=0.25):
    x = _f(logits); t = np.ascontiguousarray(targets, np.int64)
    n, c = x.shape
    out = np.empty_like(x)
    lib().oracle_sigmoid_focal_loss_fwd_f32(_p(x), _p(t), C.c_int64(n), C.c_int64(c),
                                            C.c_float(gamma), C.c_float(alpha), _p(out))
    return out


def sigmoid_focal_loss_bwd(logits, targets, d_losses, gamma=2.0, alpha=0.25):
    x = _f(logits); t = np.ascontiguousarray(targets, np.int64); g = _f(d_losses)
    n, c = x.shape
    out = np.empty_like(x)
    lib().oracle_sigmoid_focal_loss_bwd_f32(_p(x), _p(t), _p(g), C.c_int64(n), C.c_int64(c),
                                            C.c_float(gamma), C.c_float(alpha), _p(out))
    return out


def iou_loss_fwd(pred, target, eps=1e-6):
    a = _f(pred).reshape(-1, 4); b = _f(target).reshape(-1, 4)
    out = np.empty(a.shape[0], np.float32)
    lib().oracle_iou_loss_fwd_f32(_p(a), _p(b), C.c_int64(a.shape[0]), C.c_float(eps), _p(out))
    return out
