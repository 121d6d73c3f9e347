"""tests/golden/make_golden_nms_f64.py -- float64 vectors of the reference's CPU nms_ext (nms / soft_nms / nms_match on
DOUBLE tensors: AT_DISPATCH_FLOATING_TYPES evaluates them in double, nms_cpu.cpp:70,212,287), produced by the reference's own
extension compiled unmodified (oracle/build_ref.py).  Includes a case built so that the float32 and the float64 evaluation
DISAGREE (an IoU within 1e-9 of the threshold): a host routine that silently casts to float32 fails it.

    python tests/golden/make_golden_nms_f64.py            -> tests/golden/ref_nms_cpu_f64.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def synth(rng, k, w, h):
    cx, cy = rng.uniform(0, w, k), rng.uniform(0, h, k)
    bw, bh = np.exp(rng.uniform(np.log(4), np.log(320), k)), np.exp(rng.uniform(np.log(4), np.log(320), k))
    b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    s = rng.permutation(k).astype(np.float64) / k * 0.98 + 0.01 + rng.uniform(0, 1e-9, k)     # tie-free, beyond fp32 resolution
    return np.concatenate([b, s[:, None]], 1)


def main():
    from oracle import build_ref
    ext = build_ref.load_ref()
    assert ext is not None, 'reference nms extension not built (needs /root/reference)'
    rng = np.random.default_rng(97531)
    out, ci = {}, 0
    cases = [(40, 0.3, 1, 0.5, 0.05), (200, 0.45, 2, 0.3, 0.1), (200, 0.5, 0, 0.5, 0.2)]
    for k, thr, method, sigma, min_score in cases:
        d = synth(rng, k, 640, 480)
        out['dets_%d' % ci] = d
        out['params_%d' % ci] = np.array([thr, method, sigma, min_score], np.float64)
        t = torch.from_numpy(d)
        assert t.dtype == torch.float64
        out['keep_%d' % ci] = ext.nms(t, float(thr)).numpy()
        out['soft_%d' % ci] = ext.soft_nms(t, float(thr), int(method), float(sigma), float(min_score)).numpy()
        m = ext.nms_match(t, float(thr))
        out['match_sizes_%d' % ci] = np.array([len(x) for x in m], np.int64)
        out['match_members_%d' % ci] = np.array([i for x in m for i in x], np.int64)
        ci += 1
    # the discriminating case: box B overlaps box A with IoU = thr_f32 * (1 + 2e-9) in double: suppressed in double (IoU > thr),
    # while the float32 evaluation of the same rows rounds the IoU onto the threshold (not >) and keeps it
    thr = float(np.float32(0.5))
    a = np.array([0., 0., 100., 100., 0.9])
    # B = [0, 0, 100, y2]: inter = 100*100 (y2 >= 100), union = 100*y2 -> IoU = 100 / y2
    y2 = 100.0 / (thr * (1 + 2e-9))
    b = np.array([0., 0., 100., y2, 0.8])
    d = np.stack([a, b, np.array([300., 300., 320., 330., 0.7])])
    t = torch.from_numpy(d)
    k64 = ext.nms(t, thr).numpy()
    k32 = ext.nms(t.float(), thr).numpy()
    assert k64.tolist() == [0, 2] and k32.tolist() == [0, 1, 2], (k64, k32)
    out['dets_%d' % ci] = d
    out['params_%d' % ci] = np.array([thr, 1, 0.5, 1e-3], np.float64)
    out['keep_%d' % ci] = k64
    out['soft_%d' % ci] = ext.soft_nms(t, thr, 1, 0.5, 1e-3).numpy()
    m = ext.nms_match(t, thr)
    out['match_sizes_%d' % ci] = np.array([len(x) for x in m], np.int64)
    out['match_members_%d' % ci] = np.array([i for x in m for i in x], np.int64)
    ci += 1
    out['num_cases'] = np.array(ci)
    np.savez_compressed(os.path.join(HERE, 'ref_nms_cpu_f64.npz'), **out)
    print('wrote ref_nms_cpu_f64.npz: %d cases' % ci)


if __name__ == '__main__':
    main()
