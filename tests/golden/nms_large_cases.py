"""tests/golden/nms_large_cases.py -- seeded boxes of the large NMS fixtures (shared by make_golden_nms_large.py and the tests)"""
import numpy as np

CASES = [(4096, 0.3), (4096, 0.4), (8192, 0.5), (4096, 0.0)]     # (boxes, IoU threshold)


def dets(ci, W=1920, H=1080):
    """[k, 5] float32 x1, y1, x2, y2, score: centres uniform over the frame, sizes logU[4, 320], clipped; scores a permutation
    of (1 .. k) / (k + 1): tie-free"""
    k, _ = CASES[ci]
    rng = np.random.default_rng(9000 + ci)
    cx, cy = rng.uniform(0, W, k), rng.uniform(0, H, k)
    s = np.exp(rng.uniform(np.log(4), np.log(320), (k, 2)))
    b = np.stack([cx - s[:, 0] / 2, cy - s[:, 1] / 2, cx + s[:, 0] / 2, cy + s[:, 1] / 2], 1).clip(0, [W, H, W, H])
    sc = (rng.permutation(k).astype(np.float32) + 1) / (k + 1)
    return np.concatenate([b.astype(np.float32), sc[:, None].astype(np.float32)], 1)


# soft_nms / nms_match of the compiled reference (nms_cpu.cpp:76-283) on 2000 boxes in a 640 x 480 frame (dense overlaps):
# (boxes, IoU threshold, method 0 hard / 1 linear / 2 gaussian, sigma, min_score)
SOFT_CASES = [(2000, 0.3, 1, 0.5, 0.05), (2000, 0.3, 2, 0.5, 0.05), (2000, 0.5, 0, 0.5, 0.2)]


def soft_dets(ci):
    k = SOFT_CASES[ci][0]
    rng = np.random.default_rng(9100 + ci)
    W, H = 640, 480
    cx, cy = rng.uniform(0, W, k), rng.uniform(0, H, k)
    s = np.exp(rng.uniform(np.log(4), np.log(320), (k, 2)))
    b = np.stack([cx - s[:, 0] / 2, cy - s[:, 1] / 2, cx + s[:, 0] / 2, cy + s[:, 1] / 2], 1).clip(0, [W, H, W, H])
    sc = (rng.permutation(k).astype(np.float32) + 1) / (k + 1)
    return np.concatenate([b.astype(np.float32), sc[:, None].astype(np.float32)], 1)
