"""NMS / decode / fused post-processing parity (through the C ABI) -- bit-exact gates."""
import json

import numpy as np
import pytest
import torch

import oracle
from oracle import net_oracle
from conftest import load_golden, synth_boxes
from lfd_amd import configs, ops
from lfd_amd.model.utils import batched_nms, multiclass_nms, nms

pytestmark = pytest.mark.gpu


def test_nms_docstring_vector_on_device(known_answers):
    ka = known_answers['nms_docstring']
    d = torch.tensor(ka['dets'], dtype=torch.float32).cuda()
    sup, inds = nms(d, ka['iou_thr'])
    assert inds.tolist() == [0, 3, 4] and sup.shape == (3, 5)
    assert torch.equal(sup, d[inds])


def test_nms_vs_reference_extension_golden():
    g = load_golden('ref_nms.npz')
    for ci, (k, thr) in enumerate(g['cases']):
        keep = ops.nms_indices(torch.from_numpy(g['dets_%d' % ci]).cuda(), float(thr))
        assert keep.dtype == torch.long
        np.testing.assert_array_equal(keep.cpu().numpy(), g['keep_%d' % ci], err_msg='case %d' % ci)


def test_nms_vs_reference_extension_golden_at_baseline_candidate_counts():
    """device NMS == the reference's compiled nms_cpu at K = 4096 / 8192 boxes (ref_nms_large.npz; ref_nms.npz stops at 1000)"""
    import nms_large_cases as cases
    g = load_golden('ref_nms_large.npz')
    for ci, (k, thr) in enumerate(cases.CASES):
        keep = ops.nms_indices(torch.from_numpy(cases.dets(ci)).cuda(), float(thr))
        np.testing.assert_array_equal(keep.cpu().numpy(), g['keep_%d' % ci], err_msg='case %d' % ci)


@pytest.mark.parametrize('k', [1, 2, 63, 64, 65, 127, 128, 129, 1000, 4096, 9000])
@pytest.mark.parametrize('thr', [0.3, 0.4])
def test_nms_bit_exact_vs_oracle(k, thr):
    rng = np.random.default_rng(k)
    b, s = synth_boxes(rng, k)
    d = np.concatenate([b, s[:, None]], 1)
    keep = ops.nms_indices(torch.from_numpy(d).cuda(), thr).cpu().numpy()
    np.testing.assert_array_equal(keep, oracle.nms(d, thr))


def test_nms_empty_and_ties():
    e = nms(torch.zeros((0, 5)).cuda(), 0.5)
    assert e[1].numel() == 0 and e[1].dtype == torch.long
    # equal scores: contract = stable order (lower index first); identical boxes suppress each other
    d = np.array([[0, 0, 10, 10, .5], [0, 0, 10, 10, .5], [20, 20, 30, 30, .5], [0, 0, 10, 10, .9], [20, 20, 30, 30, .5]],
                 np.float32)
    keep = ops.nms_indices(torch.from_numpy(d).cuda(), 0.5).cpu().numpy()
    np.testing.assert_array_equal(keep, oracle.nms(d, 0.5))
    assert keep.tolist() == [3, 2]
    # many exact ties
    rng = np.random.default_rng(3)
    b, _ = synth_boxes(rng, 700, 640, 480)
    s = (rng.integers(0, 8, 700) / 8 + 0.0625).astype(np.float32)
    d = np.concatenate([b, s[:, None]], 1)
    np.testing.assert_array_equal(ops.nms_indices(torch.from_numpy(d).cuda(), 0.3).cpu().numpy(), oracle.nms(d, 0.3))


def test_nms_properties_at_full_size():
    """size-independent properties at K = every WF-S 1080p point (43,620): output is
    score-sorted, a subset, idempotent, and no kept pair overlaps above the threshold."""
    rng = np.random.default_rng(0)
    k = 43620
    b, s = synth_boxes(rng, k)
    d = torch.from_numpy(np.concatenate([b, s[:, None]], 1)).cuda()
    keep = ops.nms_indices(d, 0.3)
    kept = d[keep]
    assert torch.all(kept[1:, 4] <= kept[:-1, 4])
    assert keep.unique().numel() == keep.numel()
    again = ops.nms_indices(kept, 0.3)
    assert again.tolist() == list(range(kept.shape[0]))
    sub = kept[:3000]
    lt = torch.max(sub[:, None, :2], sub[None, :, :2])
    rb = torch.min(sub[:, None, 2:4], sub[None, :, 2:4])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area = (sub[:, 2] - sub[:, 0]) * (sub[:, 3] - sub[:, 1])
    iou = inter / (area[:, None] + area[None] - inter)
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.3


@pytest.mark.parametrize('agnostic', [False, True])
def test_batched_nms_offset_trick_bit_exact(agnostic):
    rng = np.random.default_rng(11)
    for k, ncls in ((5, 2), (300, 5), (3000, 45)):
        b, s = synth_boxes(rng, k, 1280, 720)
        lab = rng.integers(0, ncls, k).astype(np.int64)
        cfg = dict(type='nms', iou_thr=0.3)
        if agnostic:
            cfg['class_agnostic'] = True
        dets, keep = batched_nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(lab).cuda(), cfg)
        rd, rk = oracle.batched_nms(b, s, lab, 0.3, agnostic)
        np.testing.assert_array_equal(keep.cpu().numpy(), rk)
        np.testing.assert_array_equal(dets.cpu().numpy(), rd)     # incl. the (b+off)-off fp32 rounding


def test_multiclass_nms_vs_reference_python_golden():
    g = load_golden('ref_multiclass_nms.npz')
    for ci, (n, C, sthr, ithr, agn) in enumerate(g['cases']):
        cfg = dict(type='nms', iou_thr=float(ithr))
        if agn:
            cfg['class_agnostic'] = True
        dets, labels = multiclass_nms(torch.from_numpy(g['boxes_%d' % ci]).cuda(), torch.from_numpy(g['scores_%d' % ci]).cuda(),
                                      float(sthr), cfg)
        assert labels.device.type == 'cpu'                        # reference quirk: labels live on the CPU (nms.py:196)
        np.testing.assert_array_equal(labels.numpy(), g['labels_%d' % ci])
        np.testing.assert_array_equal(dets.cpu().numpy(), g['dets_%d' % ci])


def _model_with_sizes(name, sizes):
    m = configs.build_model(name)
    for i, s in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = tuple(int(v) for v in s)
    return m


@pytest.mark.parametrize('name', ['WIDERFACE_LFD_XS', 'WIDERFACE_LFD_S', 'TT100K_LFD_L', 'TL_LFD_L', 'TL_LFD_S'])
def test_decode_all_vs_oracle(name):
    g = load_golden('ref_model_%s.npz' % name)
    arch = configs.ARCHS[name]
    m = _model_with_sizes(name, g['sizes'])
    N, H, W = [int(v) for v in g['shape']]
    desc, P = m._detect_desc(0.5, 0.4, False)
    meta = torch.tensor([[W - 10.0, H - 6.0, 0.5]] * N, dtype=torch.float32).cuda()
    boxes, scores = ops.decode_all(desc, torch.from_numpy(g['cls']).cuda(), torch.from_numpy(g['reg']).cuda(), meta)
    strides = net_oracle.strides_of(arch)
    sizes = [tuple(s) for s in g['sizes'].tolist()]
    ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
    for n in range(N):
        rb = net_oracle.decode_boxes(g['reg'][n], sizes, strides, arch['regression_ranges'], 'sigmoid', 'union',
                                     (H - 6, W - 10), 0.5)
        rs = net_oracle.scores_from_logits(g['cls'][n], ce)
        np.testing.assert_allclose(boxes[n].cpu().numpy(), rb, rtol=1e-5, atol=2e-4)   # fp32 expf vs torch sigmoid
        np.testing.assert_allclose(scores[n].cpu().numpy(), rs, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize('name', ['WIDERFACE_LFD_XS', 'WIDERFACE_LFD_S', 'TT100K_LFD_L', 'TL_LFD_L', 'TL_LFD_S'])
def test_get_results_index_exact_vs_reference_golden(name):
    """G3: decode+NMS fed the reference's own fp32 cls/reg -> same kept detections as the
    reference's get_results (labels identical, coordinates to fp32 rounding of sigmoid)."""
    g = load_golden('ref_model_%s.npz' % name)
    m = _model_with_sizes(name, g['sizes'])
    N, H, W = [int(v) for v in g['shape']]
    m._classification_threshold = float(g['results_thr'])
    m._nms_cfg = dict(type='nms', iou_thr=float(g['results_iou']))
    cls, reg = torch.from_numpy(g['cls']).cuda(), torch.from_numpy(g['reg']).cuda()
    for key, meta in (('results', [dict(resized_height=H, resized_width=W, resize_scale=1.0)] * N),
                      ('results_scaled', [dict(resized_height=H - 6, resized_width=W - 10, resize_scale=0.5)] * N)):
        ref = json.loads(str(g[key]))
        got = m.get_results((cls, reg), meta)
        for n in range(N):
            assert len(got[n]) == len(ref[n])
            # same set of (label, box) detections; order may differ only between scores that are
            # equal up to the last ulps of expf (device) vs torch's vectorised exp (reference)
            a = np.array(got[n], np.float64).reshape(-1, 6)
            b = np.array(ref[n], np.float64).reshape(-1, 6)
            used = np.zeros(len(a), bool)
            for row in b:
                d = np.abs(a[:, 2:] - row[2:]).max(1) + 1e3 * (a[:, 0] != row[0]) + 1e3 * used
                j = int(d.argmin())
                assert d[j] < 5e-4, (row, a[j])
                assert abs(a[j, 1] - row[1]) < 1e-6 + 2e-5 * row[1]
                used[j] = True
            assert np.all(np.diff(a[:, 1]) <= 0)          # score-descending output order


def test_detect_candidate_order_and_indices_vs_oracle():
    """fused pass on oracle-decoded inputs: candidate ordinals (nonzero order), labels, point
    indices and boxes identical to the C oracle's multiclass_nms (fp16 inputs, independent decode
    so that no transcendental is involved -> fully bit-exact)."""
    rng = np.random.default_rng(5)
    sizes, strides, ranges = [(20, 30), (10, 15), (5, 8)], [8, 16, 32], ((4, 20), (20, 40), (40, 80))
    P = sum(h * w for h, w in sizes)
    for C, agn in ((1, False), (6, False), (6, True)):
        N = 3
        cls = torch.from_numpy(rng.normal(0, 2, (N, P, C)).astype(np.float32)).half()
        reg = torch.from_numpy(rng.uniform(0.05, 1.5, (N, P, 4)).astype(np.float32)).half()
        desc = ops.make_detect_desc(sizes, strides, ranges, C, C, 0, 2, agn, 4096, 0.6, 0.35)
        meta = torch.tensor([[240., 160., 1.0]] * N).cuda()
        out = ops.detect_batched(desc, cls.cuda(), reg.cuda(), meta)
        boxes, scores = ops.decode_all(desc, cls.cuda(), reg.cuda(), meta)
        counts = out.counts.cpu().numpy()
        for n in range(N):
            dets, labels, cand, K = oracle.multiclass_nms(boxes[n].cpu().numpy(), scores[n].cpu().numpy(), 0.6, 0.35, agn)
            assert counts[n, 0] == K and counts[n, 2] == 0 and counts[n, 3] == K
            k = counts[n, 1]
            assert k == len(labels)
            np.testing.assert_array_equal(out.cand[n, :k].cpu().numpy(), cand)
            np.testing.assert_array_equal(out.labels[n, :k].cpu().numpy(), labels)
            np.testing.assert_array_equal(out.dets[n, :k].cpu().numpy(), dets)
            flat = np.nonzero(scores[n].cpu().numpy().reshape(-1) > 0.6)[0]
            np.testing.assert_array_equal(out.point[n, :k].cpu().numpy(), flat[cand] // C)


def test_detect_zero_candidates_and_capacity_overflow():
    m = _model_with_sizes('WIDERFACE_LFD_S', [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)])
    P = 64 + 16 + 4 + 1 + 1
    cls = torch.full((2, P, 1), -20.0).cuda()
    reg = torch.zeros((2, P, 4)).cuda()
    meta = [dict(resized_height=64, resized_width=64, resize_scale=1.0)] * 2
    assert m.get_results((cls, reg), meta) == [[], []]
    cls[1] = 3.0                      # every point of image 1 is a candidate
    m.max_candidates = 16             # too small on purpose -> overflow flag -> exact-capacity rerun
    m._classification_threshold = 0.5
    res = m.get_results((cls, reg), meta)
    assert res[0] == [] and len(res[1]) >= 1
    sc = net_oracle.scores_from_logits(cls[1].cpu().numpy(), False)
    bx = net_oracle.decode_boxes(reg[1].cpu().numpy(), [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)], [8, 16, 32, 64, 64],
                                 configs.WIDERFACE_RANGES, 'sigmoid', 'union', (64, 64), 1.0)
    dets, labels, _, K = oracle.multiclass_nms(bx, sc, 0.5, 0.4, False)
    assert K == P and len(res[1]) == len(labels)


@pytest.mark.parametrize('C,agn,score_mode', [(1, False, 0), (6, False, 0), (6, True, 0), (5, False, 1)])
def test_detection_tail_candidate_count_sweep_vs_oracle(C, agn, score_mode):
    """Thresholds sweep the candidate count of an image from 0 through the 64-row block boundaries of the suppression mask
    up to beyond the capacity: counts, candidate ordinals and boxes against the C oracle's multiclass_nms on the
    device-decoded inputs (bit for bit), capacity overflow flagged."""
    rng = np.random.default_rng(40 + C + score_mode)
    sizes, strides, ranges = [(40, 60), (20, 30), (10, 15)], [8, 16, 32], ((4, 20), (20, 40), (40, 80))
    P = sum(h * w for h, w in sizes)
    N = 4
    Cc = C + 1 if score_mode == 1 else C
    cls = torch.from_numpy(rng.normal(0, 2, (N, P, Cc)).astype(np.float32)).half().cuda()
    reg = torch.from_numpy(rng.uniform(0.05, 1.5, (N, P, 4)).astype(np.float32)).half().cuda()
    meta = torch.tensor([[480., 320., 1.0], [480., 320., 2.0], [300., 320., 1.0], [480., 200., 0.5]]).cuda()
    sc_all = torch.sigmoid(cls.float()) if score_mode == 0 else torch.softmax(cls.float(), -1)[..., :C]
    flat = np.sort(sc_all[0].cpu().numpy().reshape(-1))[::-1]
    for target in (0, 1, 63, 64, 65, 200, 511, 512, 513, 900):
        thr = float(flat[target]) if target < flat.size else 0.0
        thr = min(thr, 0.999) if target else 1.5
        desc = ops.make_detect_desc(sizes, strides, ranges, C, Cc, score_mode, 2, agn, 512, thr, 0.35)
        a = ops.detect_batched(desc, cls, reg, meta)
        ca = a.counts.cpu().numpy()
        boxes, scores = ops.decode_all(desc, cls, reg, meta)
        for n in range(N):
            dets, labels, cand, K = oracle.multiclass_nms(boxes[n].cpu().numpy(), scores[n].cpu().numpy(), thr, 0.35, agn)
            assert ca[n, 3] == K and ca[n, 2] == int(K > 512)
            if K <= 512:
                assert ca[n, 0] == K and ca[n, 1] == len(labels)
                np.testing.assert_array_equal(a.cand[n, :len(labels)].cpu().numpy(), cand)
                np.testing.assert_array_equal(a.labels[n, :len(labels)].cpu().numpy(), labels)
                if len(labels):
                    np.testing.assert_array_equal(a.dets[n, :len(labels)].cpu().numpy(), dets)


def test_detection_tail_on_the_benchmark_grid_vs_oracle():
    """WIDERFACE_LFD_S 1080p point grid (P = 43,620), 8 frames, ~256 candidates each, clustered boxes (heavy suppression):
    the device pass against the C oracle on the device-decoded inputs, bit for bit."""
    rng = np.random.default_rng(9)
    sizes = [(135, 240), (68, 120), (34, 60), (17, 30), (17, 30)]
    strides, ranges = [8, 16, 32, 64, 64], configs.WIDERFACE_RANGES
    P = sum(h * w for h, w in sizes)
    cls = torch.from_numpy(rng.normal(-6, 1.5, (8, P, 1)).astype(np.float32))
    hot = rng.integers(0, P - 40, 30)
    for h in hot:                                  # clusters of neighbouring high-score points -> overlapping boxes
        cls[:, h:h + 9] += 7.0
    cls = cls.half().cuda()
    reg = torch.from_numpy(rng.normal(0.0, 0.4, (8, P, 4)).astype(np.float32)).half().cuda()
    meta = torch.tensor([[1920., 1080., 1.0]] * 8).cuda()
    desc = ops.make_detect_desc(sizes, strides, ranges, 1, 1, 0, 0, False, 512, 0.5, 0.4)
    a = ops.detect_batched(desc, cls, reg, meta)
    boxes, scores = ops.decode_all(desc, cls, reg, meta)
    ca = a.counts.cpu().numpy()
    assert (ca[:, 0] > 100).all() and (ca[:, 2] == 0).all() and (ca[:, 1] < ca[:, 0]).all()
    for n in range(8):
        dets, labels, cand, K = oracle.multiclass_nms(boxes[n].cpu().numpy(), scores[n].cpu().numpy(), 0.5, 0.4, False)
        k = ca[n, 1]
        assert ca[n, 0] == K and k == len(labels)
        np.testing.assert_array_equal(a.cand[n, :k].cpu().numpy(), cand)
        np.testing.assert_array_equal(a.dets[n, :k].cpu().numpy(), dets)
