"""GPU parity, sibling meta-architectures: FCOS / LFDv2 over FPN / SimpleFPN necks and 3x3 heads, through the
C ABI (lfd_detect_batched_ex, the conv / GroupNorm / upsample-add / pack kernels of engine_sibling), against the oracle
(oracle/sibling_oracle.py, pinned to the real reference in the CPU suite) and the committed reference fixtures.

Tolerances.  Post-processing is index work: kept POINT indices and labels bit-exact, boxes / scores to fp32 rounding of the
device's expf / divide (1e-5 relative).  The forward stores activations as fp16 between layers, like the LFD engine
(gate G2): logits within 2.5e-2 absolute of the reference's fp32 tensors on these O(1) logits, mean
error below 4e-3; FCOS distances (exp of a logit) within 2.5e-2 relative."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net_oracle, sibling_oracle
from conftest import load_golden
from lfd_amd import configs, ops
import sibling_cases as SC
from test_sibling_oracle_golden import NAMES, assert_results_equal, model_input, results_args

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _desc(case, thr, agn=False, cap=None):
    ce, decode, strides, ranges, pre, post = results_args(case)
    mode = {'sigmoid': 0, 'exp': 1, 'independent': 2, 'distance': 3}[decode]
    P = sum(h * w for h, w in case['sizes'])
    C = case['C']
    return ops.make_detect_desc(case['sizes'], strides, ranges, C, C + (1 if ce else 0), 1 if ce else 0, mode, agn,
                                cap or min(8192, P * C), thr, case['iou']), pre, post


@pytest.mark.parametrize('ci', range(len(SC.RESULT_CASES)))
def test_detect_ex_equals_oracle_and_reference_results(ci):
    """centerness factor + per-level top-k + decode + threshold + class-wise NMS + post-NMS cap in one device pass"""
    case = SC.RESULT_CASES[ci]
    g = load_golden('ref_sibling_results.npz')
    ref = json.loads(str(g['results_%d' % ci]))
    thr = float(g['thr_%d' % ci])
    cls, reg, ctr = SC.result_inputs(case)
    ce, decode, strides, ranges, pre, post = results_args(case)
    Hh, Ww = case['sizes'][0][0] * case['strides'][0], case['sizes'][0][1] * case['strides'][0]
    metas = [(Hh, Ww, 1.0), (Hh - 9, Ww - 14, 0.75)]
    meta = torch.tensor([[w, h, s] for h, w, s in metas], dtype=torch.float32, device=DEV)
    desc, _, _ = _desc(case, thr)
    out = ops.detect_batched_ex(desc, cls.to(DEV), reg.to(DEV), meta, centerness=None if ctr is None else ctr.to(DEV),
                                pre_nms_limit=pre, post_nms_limit=post)
    counts = out.counts.cpu().numpy()
    assert not counts[:, 2].any()
    for i, (hh, ww, sc) in enumerate(metas):
        k = int(counts[i, 1])
        dets, labels, points = sibling_oracle.get_results_single(cls[i], reg[i], None if ctr is None else ctr[i], case['sizes'],
                                                                 strides, ranges, ce, decode, thr, case['iou'], pre, post,
                                                                 (hh, ww), sc)
        assert k == len(labels)
        np.testing.assert_array_equal(out.point[i, :k].cpu().numpy(), points)          # bit-exact selection and order
        np.testing.assert_array_equal(out.labels[i, :k].cpu().numpy(), labels)
        np.testing.assert_allclose(out.dets[i, :k].cpu().numpy(), dets, rtol=1e-5, atol=1e-5)
        assert_results_equal(net_oracle.pack_results(out.dets[i, :k].cpu().numpy(), out.labels[i, :k].cpu().numpy()), ref[i])


def test_detect_ex_without_extras_equals_detect_batched():
    """no centerness, no limits: the extended entry point is the LFD pass, bit for bit"""
    case = SC.RESULT_CASES[2]
    cls, reg, _ = SC.result_inputs(case)
    meta = torch.tensor([[400., 260., 1.0], [380., 250., 0.5]], dtype=torch.float32, device=DEV)
    desc, _, _ = _desc(case, 0.9)
    a = ops.detect_batched(desc, cls.to(DEV), reg.to(DEV), meta)
    b = ops.detect_batched_ex(desc, cls.to(DEV), reg.to(DEV), meta)
    ca, cb = a.counts.cpu(), b.counts.cpu()
    assert torch.equal(ca, cb) and int(ca[:, 1].min()) > 5
    for i in range(2):
        k = int(ca[i, 1])
        assert torch.equal(a.dets[i, :k], b.dets[i, :k]) and torch.equal(a.point[i, :k], b.point[i, :k])
        assert torch.equal(a.labels[i, :k], b.labels[i, :k])


def test_pre_nms_topk_takes_the_lowest_index_among_equal_keys():
    """torch.topk leaves ties open; the device rule (and the oracle's) is lowest point index first"""
    sizes, strides = [(8, 8)], [8]
    P = 64
    cls = torch.full((1, P, 1), -2.0)
    cls[0, 10:30, 0] = 1.5          # 20 points share the best key; the limit takes 12: points 10..21
    cls[0, 40, 0] = 3.0             # and one clear winner
    reg = torch.zeros(1, P, 4)
    meta = torch.tensor([[64., 64., 1.0]], device=DEV)
    desc = ops.make_detect_desc(sizes, strides, [(0, 16)], 1, 1, 0, 1, False, 64, 0.5, 2.0)     # iou_thr 2: NMS keeps all
    out = ops.detect_batched_ex(desc, cls.to(DEV), reg.to(DEV), meta, pre_nms_limit=13)
    k = int(out.counts[0, 1])
    assert k == 13
    assert sorted(out.point[0, :k].cpu().tolist()) == [10 + i for i in range(12)] + [40]
    out = ops.detect_batched_ex(desc, cls.to(DEV), reg.to(DEV), meta, pre_nms_limit=13, post_nms_limit=4)
    assert int(out.counts[0, 1]) == 4 and int(out.point[0, 0]) == 40


def test_detect_ex_validates_arguments():
    case = SC.RESULT_CASES[4]
    cls, reg, _ = SC.result_inputs(case)
    meta = torch.ones(2, 3, device=DEV)
    desc, _, _ = _desc(case, 0.5)
    with pytest.raises(RuntimeError):
        ops.detect_batched_ex(desc, cls.to(DEV), reg.to(DEV), meta, centerness=torch.zeros(2, 7, device=DEV))
    with pytest.raises(RuntimeError):
        ops.detect_batched_ex(desc, cls, reg, meta)          # CPU tensors: no fallback


# ----------------------------------------------------------------------------------------------- element-wise operators
@pytest.mark.parametrize('shape', [((12, 16), (6, 8)), ((13, 17), (7, 9)), ((7, 9), (4, 5)), ((5, 5), (1, 1)), ((6, 8), (6, 8))])
def test_upsample_nearest_add_equals_aten(shape):
    (H, W), (h, w) = shape
    g = torch.Generator().manual_seed(3)
    dst = torch.randn(2, H, W, 64, generator=g).half()
    src = torch.randn(2, h, w, 64, generator=g).half()
    ref = dst.float().permute(0, 3, 1, 2) + F.interpolate(src.float().permute(0, 3, 1, 2), size=(H, W), mode='nearest')
    got = ops.upsample_nearest_add_(dst.to(DEV).contiguous(), src.to(DEV).contiguous())
    assert torch.equal(got.cpu(), ref.permute(0, 2, 3, 1).half())


def test_relu_maxpool_and_pack_kernels():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 9, 11, 128, generator=g).half()
    assert torch.equal(ops.relu_(x.to(DEV).clone()).cpu(), x.clamp(min=0))
    ref = F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).half()
    assert torch.equal(ops.maxpool3x3s2(x.to(DEV)).cpu(), ref)
    src = torch.randn(2, 5, 7, 32, generator=g)
    dst = torch.zeros(2, 100, 4, device=DEV)
    ops.pack_level_outputs(src.to(DEV), dst, 0, 4, 20, scale=1.25, exp=True)
    want = torch.zeros(2, 100, 4)
    want[:, 20:55] = (src[..., :4].reshape(2, 35, 4) * 1.25).exp()
    torch.testing.assert_close(dst.cpu(), want, rtol=2e-6, atol=1e-7)
    assert torch.equal(dst.cpu()[:, :20], want[:, :20]) and torch.equal(dst.cpu()[:, 55:], want[:, 55:])


# ----------------------------------------------------------------------------------------------- necks, whole models
@pytest.mark.parametrize('case', SC.NECK_CASES, ids=[c[0] for c in SC.NECK_CASES])
def test_pyramid_neck_on_device_vs_reference(case):
    name, kind, kw = case
    g = load_golden('ref_sibling_necks.npz')
    from lfd_amd.model import neck as N
    neck = getattr(N, kind)(num_input_channels_list=[64, 64, 128], num_input_strides_list=[8, 16, 32], **kw)
    configs.synthetic_weights(neck, seed=3)
    neck.eval().to(DEV)
    with torch.no_grad():
        outs = neck([t.to(DEV) for t in SC.neck_inputs(name)])
    assert len(outs) == kw['num_outputs']
    for i, y in enumerate(outs):
        ref = g['%s_out%d' % (name, i)]
        assert tuple(y.shape) == ref.shape and y.dtype == torch.float32
        err = np.abs(y.cpu().numpy() - ref)
        assert err.max() <= 2e-2 * max(1.0, np.abs(ref).max()), (i, err.max())
        assert err.mean() <= 2e-3


def _forward_on_device(name):
    g = load_golden('ref_sibling_%s.npz' % name)
    model = configs.build_sibling_model(name, seed=1).eval().to(DEV)
    with torch.no_grad():
        outs = model(model_input(g).to(DEV))
    return g, model, outs


@pytest.mark.parametrize('name', NAMES)
def test_sibling_forward_on_device_vs_reference(name):
    g, model, outs = _forward_on_device(name)
    sizes = [model.head_indexes_to_feature_map_sizes[i] for i in range(len(g['sizes']))]
    assert [list(s) for s in sizes] == g['sizes'].tolist()
    keys = ('cls', 'reg', 'ctr')[:len(outs)]
    for key, o in zip(keys, outs):
        ref = g[key]
        assert tuple(o.shape) == ref.shape and o.dtype == torch.float32
        got = o.cpu().numpy()
        # the LFD gates of tests/test_gpu_parity_fullsize.py (G2, fp16 inter-layer storage vs the reference's fp32 tensors):
        # raw logits <= 2e-2 max, <= 2e-3 mean; what decode consumes -- sigmoid(cls / centerness) -- <= 2.5e-3.  FCOSHead's
        # reg output is exp(scale * conv) (fcos_head.py:145-146): the logit error is the RELATIVE error of the distance.
        if key == 'reg' and name.startswith('FCOS'):
            err = np.abs(got - ref) / np.abs(ref)
        else:
            err = np.abs(got - ref)
        print('sibling %s %s: max %.2e mean %.2e' % (name, key, err.max(), err.mean()))
        assert err.max() <= 2e-2 and err.mean() <= 2e-3, (key, err.max(), err.mean())
        if key in ('cls', 'ctr'):
            sg = np.abs(1 / (1 + np.exp(-got.astype(np.float64))) - 1 / (1 + np.exp(-ref.astype(np.float64))))
            print('sibling %s sigmoid(%s): max %.2e' % (name, key, sg.max()))
            assert sg.max() <= 2.5e-3, (key, sg.max())


@pytest.mark.parametrize('name', NAMES)
def test_sibling_forward_replayed_from_a_hip_graph_equals_eager_launches(name):
    g = load_golden('ref_sibling_%s.npz' % name)
    model = configs.build_sibling_model(name, seed=1).eval().to(DEV)
    x = model_input(g).to(DEV)
    with torch.no_grad():
        eager = [o.clone() for o in model(x)]
        model.use_graph = True
        for _ in range(2):                       # capture, then a pure replay
            graphed = model(x)
        for a, b in zip(eager, graphed):
            assert torch.equal(a, b)


@pytest.mark.parametrize('name', NAMES)
def test_sibling_get_results_on_device_equals_oracle_on_the_same_logits(name):
    """get_results of the device model vs the oracle's get_results fed the DEVICE's forward outputs: isolates the
    post-processing (bit-exact rows) from the fp16-storage forward"""
    g, model, outs = _forward_on_device(name)
    spec = configs.SIBLINGS[name]
    ce, decode, _, ranges, pre, post = results_args(name)
    n, H, W = [int(v) for v in g['shape']]
    sizes = [tuple(s) for s in g['sizes'].tolist()]
    thr = float(g['results_thr'])
    model._classification_threshold = thr
    model._nms_cfg = dict(type='nms', iou_thr=float(g['results_iou']))
    host = [o.cpu().numpy() for o in outs]
    for hh, ww, sc in ((H, W, 1.0), (H - 6, W - 10, 0.5)):
        res = model.get_results(outs, [dict(resized_height=hh, resized_width=ww, resize_scale=sc)] * n)
        for i in range(n):
            dets, labels, _ = sibling_oracle.get_results_single(host[0][i], host[1][i], host[2][i] if len(host) == 3 else None,
                                                                sizes, list(model._point_strides), ranges, ce, decode, thr,
                                                                float(g['results_iou']), pre, post, (hh, ww), sc)
            assert len(labels) > 3
            assert_results_equal(res[i], net_oracle.pack_results(dets, labels))
    assert spec['meta'] in ('FCOS', 'LFDv2')


@pytest.mark.parametrize('name', NAMES)
def test_sibling_get_results_on_reference_logits_equals_reference_results(name):
    """the reference's own logits through the device post-processing reproduce the reference's get_results rows"""
    g = load_golden('ref_sibling_%s.npz' % name)
    model = configs.build_sibling_model(name, seed=1).eval().to(DEV)
    n, H, W = [int(v) for v in g['shape']]
    for i, hw in enumerate(g['sizes'].tolist()):
        model._head_indexes_to_feature_map_sizes[i] = tuple(hw)
    model._classification_threshold = float(g['results_thr'])
    model._nms_cfg = dict(type='nms', iou_thr=float(g['results_iou']))
    preds = tuple(torch.from_numpy(g[k]).to(DEV) for k in ('cls', 'reg', 'ctr') if k in g.files)
    for key, (hh, ww, sc) in (('results', (H, W, 1.0)), ('results_scaled', (H - 6, W - 10, 0.5))):
        res = model.get_results(preds, [dict(resized_height=hh, resized_width=ww, resize_scale=sc)] * n)
        ref = json.loads(str(g[key]))
        for i in range(n):
            assert_results_equal(res[i], ref[i])


@pytest.mark.parametrize('name', NAMES)
def test_sibling_get_loss_on_device_vs_reference(name):
    """get_loss on the reference's own predictions: loss values and prediction gradients (HIP loss kernels) vs the reference"""
    g = load_golden('ref_sibling_%s.npz' % name)
    spec = configs.SIBLINGS[name]
    model = configs.build_sibling_model(name, seed=1).to(DEV)
    n, H, W = [int(v) for v in g['shape']]
    for i, hw in enumerate(g['sizes'].tolist()):
        model._head_indexes_to_feature_map_sizes[i] = tuple(hw)
    preds = [torch.from_numpy(g[k]).to(DEV).requires_grad_(True) for k in ('cls', 'reg', 'ctr') if k in g.files]
    ann = SC.synth_annotations(5, n, H, W, spec['head']['num_classes'])
    lo = model.get_loss(tuple(preds), ann)
    ref = json.loads(str(g['loss_values']))
    assert set(lo['loss_values']) == set(ref)
    for k, v in ref.items():
        assert abs(lo['loss_values'][k] - v) <= 2e-4 * max(1.0, abs(v)), (k, lo['loss_values'][k], v)
    lo['loss'].backward()
    for nm, p in zip(('dcls', 'dreg', 'dctr'), preds):
        ref_g = g[nm]
        got = p.grad.cpu().numpy()
        scale = max(np.abs(ref_g).max(), 1e-12)
        assert np.abs(got - ref_g).max() <= 2e-4 * scale, (nm, np.abs(got - ref_g).max(), scale)


def test_fcosv1_get_loss_on_device_vs_reference():
    """FCOSv1 (fcos.py:687-768): FCOS's network with multi-label targets, flattened one-class focal loss"""
    from test_sibling_oracle_golden import _fcosv1_model_and_annotations
    g, model, ann = _fcosv1_model_and_annotations()
    ref = load_golden('ref_sibling_FCOSV1.npz')
    model.to(DEV)
    preds = [torch.from_numpy(g[k]).to(DEV).requires_grad_(True) for k in ('cls', 'reg', 'ctr')]
    lo = model.get_loss(tuple(preds), ann)
    want = json.loads(str(ref['loss_values']))
    for k, v in want.items():
        assert abs(lo['loss_values'][k] - v) <= 2e-4 * max(1.0, abs(v)), (k, lo['loss_values'][k], v)
    lo['loss'].backward()
    for nm, p in zip(('dcls', 'dreg', 'dctr'), preds):
        scale = max(np.abs(ref[nm]).max(), 1e-12)
        assert np.abs(p.grad.cpu().numpy() - ref[nm]).max() <= 2e-4 * scale, nm


def test_lfd_meta_architecture_accepts_the_sibling_modules():
    """LFD itself over an FPN neck and a 3x3 LFDHead (its constructor allows both): routed to the layer engine, same
    outputs as LFDv2 over the same modules (the two classes share forward and decode)"""
    from lfd_amd.model import LFD
    g = load_golden('ref_sibling_LFDV2_SFPN.npz')
    v2 = configs.build_sibling_model('LFDV2_SFPN', seed=1).eval().to(DEV)
    spec = configs.SIBLINGS['LFDV2_SFPN']
    v1 = LFD(backbone=v2._backbone, neck=v2._neck, head=v2._head, num_classes=spec['head']['num_classes'],
             regression_ranges=spec['regression_ranges'], range_assign_mode='dist', point_strides=v2._point_strides,
             classification_loss_func=v2._classification_loss_func, regression_loss_func=v2._regression_loss_func,
             distance_to_bbox_mode=spec['distance_to_bbox_mode']).eval()
    x = model_input(g).to(DEV)
    with torch.no_grad():
        a, b = v1(x), v2(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert [v1.head_indexes_to_feature_map_sizes[i] for i in range(4)] == [tuple(s) for s in g['sizes'].tolist()]


def test_sibling_train_mode_step_runs_through_autograd():
    """training route of the siblings (PyTorch-ROCm autograd over the same parameters + HIP loss kernels): one SGD step
    lowers the loss on the same batch.  (LFDV2_SFPN would not do: ReLU laterals + the in-place ReLU in front of its extra
    level is an autograd error in the reference as well -- the extra level rewrites a tensor ReluBackward saved.)"""
    name = 'FCOS_FPN'
    spec = configs.SIBLINGS[name]
    model = configs.build_sibling_model(name, seed=1).to(DEV).train()
    x = (torch.rand(2, 3, 128, 160, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(DEV)
    ann = SC.synth_annotations(5, 2, 128, 160, spec['head']['num_classes'])
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        lo = model.get_loss(model(x), ann)
        lo['loss'].backward()
        opt.step()
        losses.append(lo['loss_values']['loss'])
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
