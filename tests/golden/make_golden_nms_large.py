"""tests/golden/make_golden_nms_large.py -- the reference's compiled nms_cpu (nms_ext.cpp:18-27 -> cpu/nms_cpu.cpp:7-66, built
unmodified by oracle/build_ref.py) at the candidate counts of the BASELINE protocol: K = 4096 and 8192 boxes (SURVEY 8d:
centres uniform over a 1920 x 1080 frame, sizes logU[4, 320], tie-free scores).  make_golden.py's cases stop at 1000 boxes.

    python tests/golden/make_golden_nms_large.py

Also soft_nms (linear / gaussian / hard) and nms_match (nms_cpu.cpp:76-283) on 2000 densely overlapping boxes.

Output: ref_nms_large.npz -- per case the kept indices (soft_nms: the output rows; nms_match: group sizes and members); the boxes are regenerated from the seed by
the tests (make_golden.synth_boxes' recipe, restated in nms_large_cases.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import build_ref  # noqa: E402
import nms_large_cases as cases  # noqa: E402


def main():
    ext = build_ref.load_ref()
    out = {}
    for ci, (k, thr) in enumerate(cases.CASES):
        dets = cases.dets(ci)
        keep = ext.nms(torch.from_numpy(dets), float(thr)).numpy()
        out['keep_%d' % ci] = keep.astype(np.int32)
        print('K', k, 'thr', thr, 'kept', len(keep))
    for ci, (k, thr, method, sigma, min_score) in enumerate(cases.SOFT_CASES):
        dets = torch.from_numpy(cases.soft_dets(ci))
        soft = ext.soft_nms(dets, float(thr), int(method), float(sigma), float(min_score)).numpy()
        match = ext.nms_match(dets, float(thr))
        out['soft_%d' % ci] = soft
        out['match_sizes_%d' % ci] = np.array([len(m) for m in match], np.int32)
        out['match_members_%d' % ci] = np.array([i for m in match for i in m], np.int32)
        print('soft', k, thr, method, 'rows', soft.shape, 'groups', len(match))
    np.savez_compressed(os.path.join(os.environ.get('LFD_GOLDEN_OUT', HERE), 'ref_nms_large.npz'), **out)


if __name__ == '__main__':
    main()
