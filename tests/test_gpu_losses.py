"""Focal / IoU loss kernels and get_loss parity."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden
from lfd_amd import configs, ops
from lfd_amd.model.losses import FocalLoss, IoULoss
from lfd_amd.model.losses.libs import sigmoid_focal_loss_ext as ext

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,c', [(1, 1), (1000, 1), (777, 3), (4096, 45), (0, 4)])
def test_focal_forward_backward_vs_oracle(n, c):
    rng = np.random.default_rng(n + c)
    x = rng.normal(0, 3, (n, c)).astype(np.float32)
    t = rng.integers(0, c + 1, n).astype(np.int64)          # c == background
    g = rng.normal(0, 1, (n, c)).astype(np.float32)
    f = ext.forward(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), c, 2.0, 0.25)
    b = ext.backward(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(g).cuda(), c, 2.0, 0.25)
    assert f.shape == (n, c) and b.shape == (n, c)
    if n == 0:
        return
    # tolerance: device expf/logf/powf vs glibc, a few ulp of fp32
    np.testing.assert_allclose(f.cpu().numpy(), oracle.sigmoid_focal_loss_fwd(x, t), rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(b.cpu().numpy(), oracle.sigmoid_focal_loss_bwd(x, t, g), rtol=3e-5, atol=1e-6)


def test_focal_ignore_targets_and_extremes():
    x = torch.tensor([[-90., 90., 0.], [30., -30., 5.]]).cuda()
    t = torch.tensor([-1, 1]).cuda()                             # -1: ignored row (cu:33-38)
    f = ext.forward(x, t, 3, 2.0, 0.25).cpu().numpy()
    assert np.all(f[0] == 0) and np.all(np.isfinite(f))
    np.testing.assert_allclose(f, oracle.sigmoid_focal_loss_fwd(x.cpu().numpy(), t.cpu().numpy()), rtol=3e-5, atol=1e-6)


def test_focal_half_and_sum():
    rng = np.random.default_rng(2)
    x = rng.normal(0, 2, (513, 2)).astype(np.float32)
    t = rng.integers(0, 3, 513).astype(np.int64)
    ref = oracle.sigmoid_focal_loss_fwd(x.astype(np.float16).astype(np.float32), t)
    fh = ext.forward(torch.from_numpy(x).cuda().half(), torch.from_numpy(t).cuda(), 2, 2.0, 0.25)
    assert fh.dtype == torch.float16
    np.testing.assert_allclose(fh.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-4)
    s = ops.focal_sum(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), 2.0, 0.25)
    assert float(s) == pytest.approx(float(oracle.sigmoid_focal_loss_fwd(x, t).astype(np.float64).sum()), rel=1e-6)
    s2 = ops.focal_sum(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), 2.0, 0.25)
    assert float(s) == float(s2)                                  # deterministic two-stage reduction


def test_focal_module_autograd():
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.normal(0, 2, (300, 1)).astype(np.float32)).cuda().requires_grad_(True)
    t = torch.from_numpy(rng.integers(0, 2, 300).astype(np.int64)).cuda()
    loss = FocalLoss()(x, t, avg_factor=17)
    loss.backward()
    ref = oracle.sigmoid_focal_loss_fwd(x.detach().cpu().numpy(), t.cpu().numpy()).sum() / 17
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)
    gref = oracle.sigmoid_focal_loss_bwd(x.detach().cpu().numpy(), t.cpu().numpy(), np.full((300, 1), 1 / 17, np.float32))
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=3e-5, atol=1e-7)


def _ref_iou_loss(pred, target, eps=1e-6):
    """the reference expression (iou_loss.py:67-79,98-102,121-123) in torch, for autograd"""
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    a1 = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    a2 = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = torch.max(a1 + a2 - ov, ov.new_tensor([1e-6]))
    return -(ov / union).clamp(min=eps).log()


def test_iou_loss_forward_backward():
    rng = np.random.default_rng(4)
    n = 2000
    c = rng.uniform(20, 200, (n, 2))
    s1, s2 = rng.uniform(2, 60, (n, 2)), rng.uniform(2, 60, (n, 2))
    sh = rng.normal(0, 15, (n, 2))
    pred = np.concatenate([c - s1, c + s1], 1).astype(np.float32)
    tgt = np.concatenate([c + sh - s2, c + sh + s2], 1).astype(np.float32)
    pred[:5] = [[0, 0, 1, 1]] * 5
    tgt[:5] = [[50, 50, 60, 60]] * 5                               # disjoint: IoU 0 -> clamp(eps) -> zero gradient
    p = torch.from_numpy(pred).cuda().requires_grad_(True)
    loss = IoULoss()(p, torch.from_numpy(tgt).cuda(), avg_factor=n)
    loss.backward()
    np.testing.assert_allclose(ops.iou_loss_forward(p.detach(), torch.from_numpy(tgt).cuda(), 1e-6).cpu().numpy(),
                               oracle.iou_loss_fwd(pred, tgt), rtol=2e-5, atol=1e-6)
    pc = torch.from_numpy(pred).double().requires_grad_(True)
    ref = _ref_iou_loss(pc, torch.from_numpy(tgt).double()).sum() / n
    ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy(), pc.grad.numpy(), rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize('fused', ['1', '0'])
@pytest.mark.parametrize('name', ['WIDERFACE_LFD_XS', 'WIDERFACE_LFD_S', 'TT100K_LFD_L', 'TL_LFD_L', 'TL_LFD_S'])
def test_get_loss_vs_reference_golden(name, fused, monkeypatch):
    """LFD.get_loss on the reference's own predictions + annotations: loss values and gradients
    w.r.t. the predictions match the reference (its focal path ran through the C restatement).
    fused=1: the three-launch device path (csrc/getloss.hip, the default); fused=0: the op-by-op mirror."""
    monkeypatch.setenv('LFD_FUSED_LOSS', fused)
    g = load_golden('ref_model_%s.npz' % name)
    m = configs.build_model(name).cuda()
    for i, s in enumerate(g['sizes'].tolist()):
        m._head_indexes_to_feature_map_sizes[i] = tuple(s)
    cnt, o, ann = g['ann_counts'].tolist(), 0, []
    for c in cnt:
        ann.append((g['ann_boxes'][o:o + c], g['ann_labels'][o:o + c]))
        o += c
    cls = torch.from_numpy(g['cls']).cuda().requires_grad_(True)
    reg = torch.from_numpy(g['reg']).cuda().requires_grad_(True)
    out = m.get_loss((cls, reg), ann)
    assert set(out) == {'loss', 'loss_values'} and set(out['loss_values']) == {'loss', 'classification_loss', 'regression_loss'}
    lv = out['loss_values']
    np.testing.assert_allclose([lv['loss'], lv['classification_loss'], lv['regression_loss']], g['loss'], rtol=2e-5)
    out['loss'].backward()
    np.testing.assert_allclose(cls.grad.cpu().numpy(), g['grad_cls'], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(reg.grad.cpu().numpy(), g['grad_reg'], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize('fused', ['1', '0'])
def test_get_loss_no_positives(fused, monkeypatch):
    monkeypatch.setenv('LFD_FUSED_LOSS', fused)
    m = configs.build_model('WIDERFACE_LFD_XS').cuda()
    sizes = [(4, 4), (2, 2), (1, 1), (1, 1), (1, 1)]
    for i, s in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = s
    P = 16 + 4 + 3
    cls = torch.zeros(1, P, 1).cuda().requires_grad_(True)
    reg = torch.zeros(1, P, 4).cuda().requires_grad_(True)
    out = m.get_loss((cls, reg), [(np.zeros((0, 4), np.float32), np.zeros((0,), np.int64))])
    assert out['loss_values']['regression_loss'] == 0.0          # empty.sum() (lfd.py:386-387)
    out['loss'].backward()
    assert torch.isfinite(cls.grad).all()


def _random_annotations(rng, n, hw, num_classes, max_boxes):
    ann = []
    for i in range(n):
        k = int(rng.integers(0, max_boxes + 1)) if i else max_boxes      # image 0 always has boxes
        wh = np.exp(rng.uniform(np.log(6), np.log(300), (k, 2)))
        xy = rng.uniform(0, 1, (k, 2)) * (np.array([hw[1], hw[0]]) - wh).clip(1)
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), rng.integers(0, num_classes, k).astype(np.int64)))
    return ann


@pytest.mark.parametrize('name,hw,n,variant', [
    ('WIDERFACE_LFD_S', (640, 640), 4, 'plain'),
    ('WIDERFACE_LFD_S', (480, 640), 3, 'weighted'),
    ('WIDERFACE_LFD_XS', (320, 416), 2, 'exp'),
    ('TT100K_LFD_L', (384, 512), 2, 'plain'),
    ('TT100K_LFD_L', (256, 256), 2, 'weighted'),
])
def test_fused_get_loss_equals_op_by_op_path_and_oracle(name, hw, n, variant, monkeypatch):
    """The fused device get_loss (no gathers, no syncs) against (a) the op-by-op mirror of lfd.py:284-395 running the
    stand-alone HIP loss kernels and (b) the CPU oracle, at training-size point grids: losses to 2e-5 relative,
    prediction gradients to 2e-4 / 2e-3 relative (fp32 summation order differs: fp64 partials here)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32((name + variant).encode()))
    m = configs.build_model(name).cuda()
    if variant == 'weighted':
        m._enable_classification_weight = m._enable_regression_weight = True
    if variant == 'exp':
        m._distance_to_bbox_mode = 'exp'
    arch = configs.ARCHS[name]
    strides = m._point_strides
    sizes = [(-(-hw[0] // s), -(-hw[1] // s)) for s in strides]
    for i, sz in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = sz
    P = sum(h * w for h, w in sizes)
    C = m._num_classes
    ch = C + 1 if m._is_ce() else C
    ann = _random_annotations(rng, n, hw, C, 12)
    cls0 = torch.from_numpy(rng.normal(-2, 2, (n, P, ch)).astype(np.float32)).cuda()
    reg0 = torch.from_numpy(rng.normal(0, 1.0 if variant != 'exp' else 0.5, (n, P, 4)).astype(np.float32)).cuda()
    if variant == 'exp':
        reg0 += 3.0
    res = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('LFD_FUSED_LOSS', fused)
        cls, reg = cls0.clone().requires_grad_(True), reg0.clone().requires_grad_(True)
        out = m.get_loss((cls, reg), ann)
        (out['loss'] * 1.5).backward()
        res[fused] = (out['loss_values'], cls.grad.cpu().numpy(), reg.grad.cpu().numpy())
    a, b = res['1'], res['0']
    assert a[0]['regression_loss'] > 0 and np.abs(a[2]).max() > 0
    for k in ('loss', 'classification_loss', 'regression_loss'):
        assert a[0][k] == pytest.approx(b[0][k], rel=2e-5), k
    np.testing.assert_allclose(a[1], b[1], rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(a[2], b[2], rtol=2e-3, atol=1e-9)
    # gray and non-positive rows get exactly zero regression gradient, gray rows exactly zero class gradient
    assert np.array_equal(a[2] == 0, b[2] == 0) or np.abs(a[2][(a[2] == 0) != (b[2] == 0)]).max() < 1e-12
    if variant == 'plain':
        import oracle.net_oracle as no
        o = no.lfd_loss(dict(arch, distance_to_bbox_mode=m._distance_to_bbox_mode), cls0.cpu(), reg0.cpu(), sizes,
                        list(strides), [torch.from_numpy(x) for x, _ in ann], [torch.from_numpy(y) for _, y in ann])
        assert a[0]['loss'] == pytest.approx(o['loss'], rel=2e-5)
        assert a[0]['classification_loss'] == pytest.approx(o['classification_loss'], rel=2e-5)
        assert a[0]['regression_loss'] == pytest.approx(o['regression_loss'], rel=2e-5)


def test_fused_get_loss_sums_are_deterministic_and_global_normaliser():
    """Same inputs -> bit-identical sums (fixed-order fp64 reduction); the finalize stage divides by the GLOBAL sums
    handed in (image-parallel ranks): feeding 2x the local sums as 'global' with rank_scale 2 halves nothing but the
    +1 of the classification normaliser."""
    rng = np.random.default_rng(5)
    sizes, strides, ranges = [(40, 40), (20, 20)], [8, 16], [(4, 20), (20, 40)]
    P, n, C = 2000, 3, 1
    d = ops.make_loss_desc(n, sizes, strides, ranges, C, False, 'sigmoid')
    ct = torch.from_numpy(np.where(rng.random((n, P, C)) < 0.05, rng.random((n, P, C)), 0).astype(np.float32)).cuda()
    ct[0, :50] = -1
    rt = torch.from_numpy(rng.uniform(1, 30, (n, P, 4)).astype(np.float32)).cuda()
    pc = torch.from_numpy(rng.normal(0, 2, (n, P, C)).astype(np.float32)).cuda()
    pr = torch.from_numpy(rng.normal(0, 1, (n, P, 4)).astype(np.float32)).cuda()
    f1 = ops.get_loss_forward(d, pc, pr, ct, rt)
    f2 = ops.get_loss_forward(d, pc, pr, ct, rt)
    assert torch.equal(f1, f2)
    n_pos = float(f1[3])
    assert n_pos == float(((ct.max(-1)[0] >= 0.001) & (ct.min(-1)[0] >= 0)).sum())
    assert float(f1[6]) == float((ct.min(-1)[0] >= 0).sum())
    g = ops.get_loss_forward(d, pc, pr, ct, rt, reduce_sums=lambda s: s * 2, rank_scale=2.0)
    assert float(g[3]) == 2 * n_pos
    assert float(g[1]) == pytest.approx(float(f1[1]), rel=1e-6)                       # 2 * sum / (2 n_pos)
    assert float(g[0]) == pytest.approx(float(f1[0]) * (n_pos + 1) * 2 / (2 * n_pos + 1), rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['WIDERFACE_LFD_XS', 'WIDERFACE_LFD_S', 'TT100K_LFD_L', 'TL_LFD_L', 'TL_LFD_S'])
def test_device_target_assignment_bit_exact_vs_reference_golden(name):
    """lfd_assign_targets_f32 (csrc/targets.hip) == the reference's annotation_to_target output (golden, generated by
    the real reference on the CPU), bit for bit: cls targets incl. gray (-1) cells, and reg targets of EVERY point
    (also the don't-care rows, which depend on the sort-order details of lfd.py:230-249)."""
    g = load_golden('ref_model_%s.npz' % name)
    m = configs.build_model(name).cuda()
    for i, s in enumerate(g['sizes'].tolist()):
        m._head_indexes_to_feature_map_sizes[i] = tuple(s)
    pts = m.generate_point_coordinates(m.head_indexes_to_feature_map_sizes)
    cnt, o, bb, ll = g['ann_counts'].tolist(), 0, [], []
    for c in cnt:
        bb.append(torch.from_numpy(g['ann_boxes'][o:o + c]).cuda())
        ll.append(torch.from_numpy(g['ann_labels'][o:o + c]).cuda())
        o += c
    ct, rt = m.annotation_to_target(pts, bb, ll)
    assert ct.is_cuda and ct.dtype == torch.float32
    np.testing.assert_array_equal(ct.cpu().numpy(), g['cls_targets'])
    np.testing.assert_array_equal(rt.cpu().numpy(), g['reg_targets'])


@pytest.mark.gpu
@pytest.mark.parametrize('mode,independent', [('longer', False), ('shorter', True), ('sqrt', False), ('dist', True)])
def test_device_target_assignment_modes_vs_oracle(mode, independent):
    """all four range_assign_mode values, both regression conventions, duplicated / overlapping boxes (ties), an
    image without annotations, > 128 boxes in one image (the kernel stages boxes in LDS chunks of 128)."""
    from lfd_amd import ops
    from oracle import net_oracle
    rng = np.random.default_rng(7)
    sizes, strides = [(12, 20), (6, 10), (3, 5)], [4, 8, 16]
    rr, gr = [(4, 20), (20, 40), (40, 80)], [(3, 22), (18, 44), (36, 88)]
    imgs = []
    for n_box in (150, 0, 5):
        xy = rng.uniform(0, 70, size=(n_box, 2)).astype(np.float32)
        wh = rng.uniform(3, 70, size=(n_box, 2)).astype(np.float32).round()
        b = np.concatenate([xy.round(), wh], 1).astype(np.float32)
        if n_box >= 5:
            b[1] = b[0]                 # exact duplicate -> equal scores
            b[3, :2] = b[2, :2]         # same corner, different size
        imgs.append((b, rng.integers(0, 3, size=(n_box,)).astype(np.int64)))
    ct, rt = ops.assign_targets(sizes, strides, rr, gr, 3, mode, independent,
                                [torch.from_numpy(b).cuda() for b, _ in imgs], [torch.from_numpy(l).cuda() for _, l in imgs])
    pts = torch.from_numpy(np.concatenate(net_oracle.point_coordinates(sizes, strides), 0))
    st = torch.cat([torch.full((h * w,), s, dtype=torch.int64) for (h, w), s in zip(sizes, strides)])
    rrp = torch.cat([torch.tensor(r, dtype=torch.int64)[None].expand(h * w, 2) for (h, w), r in zip(sizes, rr)])
    grp = torch.cat([torch.tensor(r, dtype=torch.int64)[None].expand(h * w, 2) for (h, w), r in zip(sizes, gr)])
    for i, (b, l) in enumerate(imgs):
        ec, er = net_oracle.assign_targets_single(pts, st, rrp, grp, torch.from_numpy(b), torch.from_numpy(l), 3,
                                                  range_assign_mode=mode, loss_type='independent' if independent else 'union')
        # decisions (gray = -1, negative = 0, which box supplies the regression target) must be identical.  The score
        # values go through sqrt(1/x): the kernel evaluates it with IEEE divide / sqrt, torch's CPU sqrt in this
        # build is a <= 1 ulp routine (1388 of 200k random inputs differ from numpy's IEEE result), so scores are
        # compared to one ulp of a value <= 1 (1.2e-7); on the reference-generated golden fixtures they are equal.
        got, exp = ct[i].cpu().numpy(), ec.numpy()
        np.testing.assert_array_equal(got == -1, exp == -1)
        np.testing.assert_array_equal(got == 0, exp == 0)
        np.testing.assert_allclose(got, exp, rtol=0, atol=1.2e-7)
        # Regression targets: lfd.py:230-249 picks "the first maximum of the filtered scores in ascending-sorted
        # order"; torch's CPU sort is not stable, so among boxes with EQUAL scores (several boxes whose centre is
        # within half a stride of the point all score exactly 1.0) the reference's pick is implementation defined.
        # The kernel's rule (lowest box index) must be one of the admissible picks; where the score is unique it
        # must be the oracle's pick.
        pxy = pts.numpy().astype(np.float32)
        hs = (st.numpy().astype(np.float32) / np.float32(2))[:, None]
        G = b.shape[0]
        got_r, exp_r = rt[i].cpu().numpy(), er.numpy()
        if G == 0:
            np.testing.assert_array_equal(got_r, exp_r)
            continue
        bx, by, bw, bh = (b[None, :, k] for k in range(4))
        def axis(d):
            v = (d / hs).astype(np.float32)
            v = (v * (v >= 1) + (v < 1)).astype(np.float32)
            return np.sqrt((np.float32(1) / v).astype(np.float32)).astype(np.float32)
        score = (axis(np.abs(pxy[:, :1] - (bx + bw / np.float32(2)))) * axis(np.abs(pxy[:, 1:2] - (by + bh / np.float32(2))))).astype(np.float32)
        delta = np.stack([pxy[:, :1] - bx, pxy[:, 1:2] - by, (bx + bw - np.float32(1)) - pxy[:, :1], (by + bh - np.float32(1)) - pxy[:, 1:2]], -1).astype(np.float32)
        measure = {'longer': np.maximum(bw, bh) + 0 * score, 'shorter': np.minimum(bw, bh) + 0 * score,
                   'sqrt': np.sqrt(bw * bh).astype(np.float32) + 0 * score, 'dist': delta.max(-1)}[mode]
        rlo, rhi = rrp.numpy().astype(np.float32)[:, :1], rrp.numpy().astype(np.float32)[:, 1:2]
        if independent:
            delta = (delta / rhi[..., None]).astype(np.float32)
        green = (rlo <= measure) & (measure <= rhi) & (delta.min(-1) >= 0)
        filt = score * green
        fmax = filt.max(1, keepdims=True)
        cand = np.where(fmax > 0, filt == fmax, score == score.min(1, keepdims=True))       # admissible picks
        ok = ((delta == got_r[:, None, :]).all(-1) & cand).any(1)
        assert ok.all(), int((~ok).sum())
        # "unique" with a margin of a few ulp: torch's <= 1 ulp sqrt can reorder two scores that differ in the last bit
        key = np.where(fmax > 0, filt, -score)
        srt = np.sort(key, 1)
        margin = (srt[:, -1] - srt[:, -2]) if G > 1 else np.ones(len(key), np.float32)
        unique = margin > 4e-7
        assert unique.mean() > 0.5
        np.testing.assert_array_equal(got_r[unique], exp_r[unique])


@pytest.mark.gpu
def test_cross_entropy_kernel_vs_torch():
    """lfd_cross_entropy_{fwd,bwd}_f32 vs F.cross_entropy(reduction='none') in fp64 (46 classes = TT100K + background)."""
    from lfd_amd.model.losses import CrossEntropyLoss
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1000, 46, generator=g) * 4)
    t = torch.randint(0, 46, (1000,), generator=g)
    w = torch.rand(1000, generator=g)
    xr = x.double().requires_grad_(True)
    ref = (torch.nn.functional.cross_entropy(xr, t, reduction='none') * w.double()).sum() / 37.0
    ref.backward()
    xg = x.cuda().requires_grad_(True)
    out = CrossEntropyLoss(reduction='mean', loss_weight=1.0)(xg, t.cuda(), weight=w.cuda(), avg_factor=37.0)
    out.backward()
    assert float(out) == pytest.approx(float(ref), rel=2e-6)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.float().numpy(), rtol=2e-5, atol=1e-8)


def _ref_union_losses(pred, target, kind, eps):
    from oracle import net_oracle
    return net_oracle.union_box_loss(pred, target, kind, eps)


@pytest.mark.parametrize('kind', ['giou', 'diou', 'ciou'])
def test_box_loss_kernel_vs_reference_golden(kind):
    """lfd_box_loss_f32 on the reference-generated vectors (tests/golden/ref_box_losses.npz): loss and d(sum loss)/d(pred)
    of the real reference modules, both computed in fp32 -- agreement to fp32 rounding of the same expression."""
    g = load_golden('ref_box_losses.npz')
    loss, grad = ops.box_loss(torch.from_numpy(g['pred']).cuda(), torch.from_numpy(g['target']).cuda(), kind, float(g['eps']))
    np.testing.assert_allclose(loss.cpu().numpy(), g['loss_' + kind], rtol=5e-5, atol=5e-6)
    gk, gr = grad.cpu().numpy(), g['grad_' + kind]
    bad = np.abs(gk - gr) > 1e-3 * np.abs(gr) + 1e-5 * np.abs(gr).max()
    assert bad.mean() < 2e-3, bad.mean()


@pytest.mark.parametrize('kind', ['giou', 'diou', 'ciou'])
def test_giou_diou_ciou_losses_and_gradients(kind):
    """lfd_box_loss_f32 (gradient by forward-mode differentiation inside the kernel) vs autograd of the same expression in
    float64: overlapping, disjoint, nested and nearly identical pairs; then the loss modules (weights, avg_factor)."""
    from lfd_amd.model.losses import CIoULoss, DIoULoss, GIoULoss
    rng = np.random.default_rng({'giou': 1, 'diou': 2, 'ciou': 3}[kind])
    n = 4096
    c = rng.uniform(20, 400, (n, 2)); s = np.exp(rng.uniform(np.log(4), np.log(200), (n, 2)))
    tgt = np.concatenate([c - s / 2, c + s / 2], 1)
    shift = rng.normal(0, 1, (n, 2)) * s * rng.choice([0.05, 0.5, 3.0], (n, 1))      # near, overlapping, disjoint
    ps = s * np.exp(rng.normal(0, 0.5, (n, 2)))
    pred = np.concatenate([c + shift - ps / 2, c + shift + ps / 2], 1)
    pred[:64] = tgt[:64] + rng.normal(0, 1e-3, (64, 4))                                # nearly identical
    pred[64:128, :2] = tgt[64:128, :2] + 1.0; pred[64:128, 2:] = tgt[64:128, 2:] - 1.0  # nested
    pred, tgt = pred.astype(np.float32), tgt.astype(np.float32)
    p64 = torch.from_numpy(pred).double().requires_grad_(True)
    ref = _ref_union_losses(p64, torch.from_numpy(tgt).double(), kind, 1e-7)
    g = torch.from_numpy(rng.normal(0, 1, n)).double()
    (ref * g).sum().backward()
    loss, grad = ops.box_loss(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), kind, 1e-7)
    np.testing.assert_allclose(loss.cpu().numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    gk = (grad.cpu().double() * g[:, None]).numpy()
    gr = p64.grad.numpy()
    bad = np.abs(gk - gr) > 2e-3 * np.abs(gr) + 2e-5 * np.abs(gr).max()
    assert bad.mean() < 2e-3, bad.mean()          # fp32 kernel vs fp64 autograd; a few ill-conditioned (near-tie) pairs allowed
    mod = {'giou': GIoULoss, 'diou': DIoULoss, 'ciou': CIoULoss}[kind](eps=1e-6, loss_weight=2.0)
    pred, tgt = pred[128:], tgt[128:]           # the module check skips the ill-conditioned (nearly identical / nested) pairs
    pc = torch.from_numpy(pred[:300]).cuda().requires_grad_(True)
    w = torch.from_numpy(rng.uniform(0, 1, 300).astype(np.float32)).cuda()
    out = mod(pc, torch.from_numpy(tgt[:300]).cuda(), weight=w, avg_factor=37.0)
    out.backward()
    pr = torch.from_numpy(pred[:300]).double().requires_grad_(True)
    ref2 = 2.0 * (_ref_union_losses(pr, torch.from_numpy(tgt[:300]).double(), kind, 1e-6) * w.cpu().double()).sum() / 37.0
    ref2.backward()
    assert float(out.detach()) == pytest.approx(float(ref2.detach()), rel=1e-4)
    assert float((pc.grad.cpu().double() - pr.grad).abs().max()) < 2e-3 * float(pr.grad.abs().max())
    zero = mod(pc, torch.from_numpy(tgt[:300]).cuda(), weight=torch.zeros(300, 4).cuda())
    assert float(zero) == 0.0          # all-zero weights: (pred * weight).sum() (iou_loss.py:339-340; [n,4] weights broadcast)


def test_get_loss_with_a_giou_regression_loss_runs_the_op_by_op_path():
    """LFD accepts the whole IoU-loss family (lfd.py:64-66); only IoULoss has the fused get_loss kernels, the others go
    through the op-by-op mirror with the box-loss kernel behind the module."""
    from lfd_amd.model.losses import GIoULoss
    m = configs.build_model('WIDERFACE_LFD_XS').cuda()
    m._regression_loss_func = GIoULoss(eps=1e-6, reduction='mean', loss_weight=1.0)
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    for i, s in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = s
    P = sum(h * w for h, w in sizes)
    g = torch.Generator(device='cuda').manual_seed(0)
    cls = torch.randn(2, P, 1, generator=g, device='cuda').requires_grad_(True)
    reg = torch.randn(2, P, 4, generator=g, device='cuda').requires_grad_(True)
    ann = [(np.array([[8., 8., 20., 24.]], np.float32), np.zeros(1, np.int64)),
           (np.array([[20., 10., 30., 30.]], np.float32), np.zeros(1, np.int64))]
    assert not m._fused_loss_supported(cls)
    out = m.get_loss((cls, reg), ann)
    assert out['loss_values']['regression_loss'] > 0
    out['loss'].backward()
    assert torch.isfinite(reg.grad).all() and float(reg.grad.abs().sum()) > 0


@pytest.mark.parametrize('key,kind,beta', [('smooth_l1', 'smooth_l1', 1.0), ('smooth_l1_b011', 'smooth_l1', 0.11),
                                           ('l1', 'l1', 1.0), ('mse', 'mse', 1.0)])
def test_pointwise_regression_losses_vs_reference_golden(key, kind, beta):
    """lfd_pointwise_loss_f32 (SmoothL1 / L1 / MSE, LFD's "independent" regression losses) on the reference-generated
    vectors: loss and derivative, incl. exact zeros (sign(0) = 0 like torch.abs) -- fp32 expressions of the same shape, so
    agreement to 1 ulp; then the modules with weight / avg_factor against the oracle."""
    from lfd_amd.model.losses import L1Loss, MSELoss, SmoothL1Loss
    from oracle import net_oracle
    g = load_golden('ref_box_losses.npz')
    a, b = torch.from_numpy(g['pw_pred']).cuda(), torch.from_numpy(g['pw_target']).cuda()
    loss, grad = ops.pointwise_loss(a, b, kind, beta)
    np.testing.assert_allclose(loss.cpu().numpy(), g['loss_' + key], rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(grad.cpu().numpy(), g['grad_' + key], rtol=3e-7, atol=1e-9)
    assert float(grad[:16].abs().max()) == 0.0
    mod = {'smooth_l1': SmoothL1Loss(beta=beta, loss_weight=0.5), 'l1': L1Loss(loss_weight=0.5), 'mse': MSELoss(loss_weight=0.5)}[kind]
    ar = a.clone().requires_grad_(True)
    w = torch.linspace(0.1, 2.0, a.size(0), device='cuda')[:, None].expand_as(a)
    out = mod(ar, b, weight=w, avg_factor=11.0)
    out.backward()
    ac = a.cpu().double().requires_grad_(True)
    ref = 0.5 * (net_oracle.pointwise_reg_loss(ac, b.cpu().double(), kind, beta) * w.cpu().double()).sum() / 11.0
    ref.backward()
    assert float(out.detach()) == pytest.approx(float(ref.detach()), rel=1e-5)
    np.testing.assert_allclose(ar.grad.cpu().numpy(), ac.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_get_loss_independent_regression_mode_runs_on_the_device():
    """LFD with a SmoothL1Loss regression loss ('independent' family, lfd.py:61-66,354-358): targets in independent mode
    from the device assignment kernel, regression loss on the raw predictions."""
    from lfd_amd.model.losses import SmoothL1Loss
    arch = dict(configs.ARCHS['WIDERFACE_LFD_XS'])
    m = configs.build_model(arch).cuda()
    m._regression_loss_func = SmoothL1Loss(beta=1.0)
    m._regression_loss_type = 'independent'
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    for i, s in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = s
    P = sum(h * w for h, w in sizes)
    gen = torch.Generator(device='cuda').manual_seed(0)
    cls = torch.randn(2, P, 1, generator=gen, device='cuda').requires_grad_(True)
    reg = torch.randn(2, P, 4, generator=gen, device='cuda').requires_grad_(True)
    ann = [(np.array([[8., 8., 20., 24.]], np.float32), np.zeros(1, np.int64)),
           (np.array([[20., 10., 30., 30.]], np.float32), np.zeros(1, np.int64))]
    out = m.get_loss((cls, reg), ann)
    assert out['loss_values']['regression_loss'] > 0
    out['loss'].backward()
    assert torch.isfinite(reg.grad).all() and float(reg.grad.abs().sum()) > 0


def test_bce_with_logits_and_quality_focal_loss_vs_reference_golden():
    """lfd_bce_with_logits_f32 / lfd_quality_focal_loss_f32 through the host modules (BCEWithLogitsLoss with float targets
    and with 1-based labels + row weights; QualityFocalLoss) on the reference-generated vectors: per-element / per-row loss
    and gradient; then LFD.get_loss with a QualityFocalLoss classification loss runs on the device."""
    from lfd_amd.model.losses import BCEWithLogitsLoss, QualityFocalLoss
    g = load_golden('ref_box_losses.npz')
    x = torch.from_numpy(g['cls_logits']).cuda()

    def check(key, fn):
        p = x.clone().requires_grad_(True)
        loss = fn(p)
        loss.sum().backward()
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g['loss_' + key], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(p.grad.cpu().numpy(), g['grad_' + key], rtol=1e-4, atol=2e-6)
    check('bce_soft', lambda p: BCEWithLogitsLoss(reduction='none')(p, torch.from_numpy(g['bce_soft']).cuda()))
    check('bce_labels', lambda p: BCEWithLogitsLoss(reduction='none')(p, torch.from_numpy(g['bce_labels']).cuda(),
                                                                      weight=torch.from_numpy(g['bce_row_weight']).cuda()))
    check('qfl', lambda p: QualityFocalLoss(beta=2.0, reduction='none')(
        p, (torch.from_numpy(g['qfl_labels']).cuda(), torch.from_numpy(g['qfl_scores']).cuda())))
    m = configs.build_model('WIDERFACE_LFD_XS').cuda()
    m._classification_loss_func = QualityFocalLoss(beta=2.0)
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    for i, sz in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = sz
    P = sum(h * w for h, w in sizes)
    gen = torch.Generator(device='cuda').manual_seed(0)
    cls = torch.randn(2, P, 1, generator=gen, device='cuda').requires_grad_(True)
    reg = torch.randn(2, P, 4, generator=gen, device='cuda').requires_grad_(True)
    ann = [(np.array([[8., 8., 20., 24.]], np.float32), np.zeros(1, np.int64)),
           (np.array([[20., 10., 30., 30.]], np.float32), np.zeros(1, np.int64))]
    out = m.get_loss((cls, reg), ann)
    assert out['loss_values']['classification_loss'] > 0
    out['loss'].backward()
    assert torch.isfinite(cls.grad).all() and float(cls.grad.abs().sum()) > 0


def test_image_parallel_normalisers_fused_and_op_by_op_paths_agree(monkeypatch):
    """Simulated world of 2 (lfd_amd.parallel patched: the "other rank" contributes 7 positives with score sum 3.5): both
    get_loss paths must use the GLOBAL n_pos (+1) / score-sum normalisers and scale the rank's loss by the world size --
    and therefore agree with each other and with the hand-computed rescaling of the single-process loss."""
    from lfd_amd import parallel
    m = configs.build_model('WIDERFACE_LFD_S').cuda()
    sizes = [(20, 24), (10, 12), (5, 6), (3, 3), (3, 3)]
    for i, sz in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = sz
    P = sum(h * w for h, w in sizes)
    rng = np.random.default_rng(3)
    ann = _random_annotations(rng, 2, (160, 192), 1, 6)
    cls0 = torch.from_numpy(rng.normal(-2, 2, (2, P, 1)).astype(np.float32)).cuda()
    reg0 = torch.from_numpy(rng.normal(0, 1, (2, P, 4)).astype(np.float32)).cuda()

    def run(fused):
        monkeypatch.setenv('LFD_FUSED_LOSS', fused)
        c, r = cls0.clone().requires_grad_(True), reg0.clone().requires_grad_(True)
        out = m.get_loss((c, r), ann)
        out['loss'].backward()
        return out['loss_values'], c.grad.clone(), r.grad.clone()
    single = run('1')
    other_npos, other_w = 7.0, 3.5

    def fake_count(t):
        add = torch.zeros_like(t)
        if t.numel() == 8:            # fused path: {cls_sum, reg_sum, n_pos, score_sum, n_green, ...}
            add[2], add[3] = other_npos, other_w
        else:                         # op-by-op path: {n_pos, score_sum}
            add[0], add[1] = other_npos, other_w
        return t + add
    monkeypatch.setattr(parallel, 'is_dist', lambda: True)
    monkeypatch.setattr(parallel, 'world_size', lambda: 2)
    monkeypatch.setattr(parallel, 'global_count', fake_count)
    a, b = run('1'), run('0')
    for k in ('loss', 'classification_loss', 'regression_loss'):
        assert a[0][k] == pytest.approx(b[0][k], rel=2e-5), k
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(a[2].cpu().numpy(), b[2].cpu().numpy(), rtol=2e-3, atol=1e-9)
    # cls_single = S / (n + 1), cls_dist = 2 S / (n + 7 + 1): solve for the local positive count n
    cs, cd = single[0]['classification_loss'], a[0]['classification_loss']
    n = (2 * cs - 8 * cd) / (cd - 2 * cs) if abs(cd - 2 * cs) > 1e-12 else 0.0   # from cd (n + 8) = 2 cs (n + 1)
    assert n > 0 and abs(n - round(n)) < 1e-2, n                                 # an integer number of local positives
    rs, rd = single[0]['regression_loss'], a[0]['regression_loss']
    assert rd == pytest.approx(2 * rs * round(n) / (round(n) + 7), rel=1e-4)
