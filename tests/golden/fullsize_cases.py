"""tests/golden/fullsize_cases.py -- seeded synthetic logits on the FULL point grids of BASELINE configs 2 / 3 / 4, shared by the
generator (make_golden_fullsize_results.py, runs the reference's get_results) and the tests that read its fixture."""
import numpy as np

# key -> (configuration, (H, W) of the frame, per-level (h, w) of the point grid as LFD.forward leaves them in
# head_indexes_to_feature_map_sizes at that frame size; SURVEY 8 [probe]: P = 43,620 / 690,600 / 76,520)
GRIDS = {
    'config2': ('WIDERFACE_LFD_S', (1080, 1920), [(135, 240), (68, 120), (34, 60), (17, 30), (17, 30)]),
    'config3': ('WIDERFACE_LFD_L', (2160, 3840), [(540, 960), (270, 480), (135, 240), (68, 120), (34, 60)]),
    'config4': ('TT100K_LFD_L', (720, 1280), [(180, 320), (90, 160), (45, 80), (23, 40)]),
}
# (candidates K the threshold is set for, IoU threshold, class_agnostic, resize_scale)
SETTINGS = [(256, 0.4, False, 1.0), (4096, 0.4, False, 1.0), (4096, 0.3, True, 0.5)]


def logits(key, channels):
    """fp32 logits, tie-free for all practical purposes (continuous draws, no fp16 rounding): cls [1, P, channels] with a
    thin upper tail, reg [1, P, 4]"""
    _, _, sizes = GRIDS[key]
    p = sum(h * w for h, w in sizes)
    rs = np.random.default_rng({'config2': 22, 'config3': 33, 'config4': 44}[key])
    cls = rs.normal(-4.0, 1.6, (1, p, channels)).astype(np.float32)
    reg = rs.normal(0.0, 1.2, (1, p, 4)).astype(np.float32)
    return cls, reg


def frame(key):
    """one seeded frame [1, 3, H, W] in [-1, 1) (fp32) for the forward fixtures"""
    import torch
    _, (h, w), _ = GRIDS[key]
    return torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed({'config2': 2, 'config3': 3, 'config4': 4}[key])) * 2 - 1


def sample_index(p):
    """every 37th point (37 is coprime to every level's width): a strided sample that visits all levels and columns"""
    return np.arange(0, p, 37)


# BASELINE config 5: WIDERFACE_LFD_S trained on 640 x 640 crops, 32 per GPU, G ~ U{1..20} boxes per image with
# w, h ~ logU[6, 300] clipped to the frame, so that every level's range (4 .. 320 px) gets positives
TRAIN = ('WIDERFACE_LFD_S', 32, 640, 640, [(80, 80), (40, 40), (20, 20), (10, 10), (10, 10)])


def train_annotations():
    name, n, h, w, _ = TRAIN
    rs = np.random.default_rng(55)
    ann = []
    for _ in range(n):
        g = int(rs.integers(1, 21))
        wh = np.minimum(np.exp(rs.uniform(np.log(6), np.log(300), (g, 2))), [w, h])
        xy = rs.uniform(0, [w, h], (g, 2)) - wh / 2
        xy = np.clip(xy, 0, np.array([w, h]) - wh)
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), np.zeros(g, np.int64)))
    return ann


def train_logits():
    name, n, h, w, sizes = TRAIN
    p = sum(a * b for a, b in sizes)
    rs = np.random.default_rng(56)
    return rs.normal(-4.0, 1.6, (n, p, 1)).astype(np.float32), rs.normal(0.0, 1.2, (n, p, 4)).astype(np.float32)
