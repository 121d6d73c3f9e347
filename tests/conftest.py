"""Test configuration.  `-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI exports
(no compute calls).  `-m gpu` : parity of the HIP path (through the C ABI) against the oracle
and the committed golden fixtures; runs on the MI355X box, where /root/reference does not exist."""
import json
import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)      # sibling_cases.py: seeded inputs shared with the fixture generator
warnings.filterwarnings('ignore', message='.*indexing.*')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible in this process')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def known_answers():
    return json.load(open(os.path.join(GOLDEN, 'known_answers.json')))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_boxes(rng, k, W=1920, H=1080):
    """SURVEY 8d synthetic NMS load: centres uniform, sizes logU[4,320], tie-free scores."""
    cx, cy = rng.uniform(0, W, k), rng.uniform(0, H, k)
    s = np.exp(rng.uniform(np.log(4), np.log(320), (k, 2)))
    b = np.stack([cx - s[:, 0] / 2, cy - s[:, 1] / 2, cx + s[:, 0] / 2, cy + s[:, 1] / 2], 1)
    b = b.clip(0, [W, H, W, H]).astype(np.float32)
    sc = (rng.permutation(k).astype(np.float32) + 1) / (k + 1)
    return b, sc.astype(np.float32)
