"""CPU suite, sibling meta-architectures: the oracle restatement (oracle/sibling_oracle.py) and the host mirror
(lfd_amd.model.{fcos,lfdv2,neck,head}) against fixtures produced by the REAL reference classes
(tests/golden/make_golden_siblings.py)."""
import hashlib
import json

import numpy as np
import pytest
import torch

from oracle import net_oracle, sibling_oracle
from conftest import load_golden
from lfd_amd import configs
import sibling_cases as SC

NAMES = sorted(configs.SIBLINGS)


def _sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def oracle_neck(spec):
    nk = spec['neck']
    if nk['kind'] == 'SimpleNeck':
        return None
    norm = None
    if nk.get('norm_on_lateral'):
        c = nk['norm_cfg']
        norm = 'BatchNorm2d' if c['type'] == 'BatchNorm2d' else ('GroupNorm', c['num_groups'])
    return dict(kind=nk['kind'], num_inputs=3, num_outputs=nk['num_outputs'], extra_on_input=nk.get('extra_on_input', False),
                extra_type=nk.get('extra_type', 'conv'), norm_on_lateral=norm, relu_on_lateral=nk.get('relu_on_lateral', False),
                relu_before_extra=nk.get('relu_before_extra', nk['kind'] == 'SimpleFPN'),
                neighbouring_mode=nk.get('neighbouring_mode', False))


def oracle_arch(spec):
    """net_oracle-style arch dict of a SIBLINGS entry (backbone + LFDHead kwargs)"""
    a = dict(spec['backbone'])
    hk = spec['head']
    if hk['kind'] == 'LFDHeadV1':
        a.update(classification_loss_type=spec['classification_loss_type'], regression_loss_type=spec['regression_loss_type'],
                 regression_ranges=spec['regression_ranges'], distance_to_bbox_mode=spec['distance_to_bbox_mode'])
    if hk['kind'] == 'LFDHead':
        a.update(num_classes=hk['num_classes'], num_head_channels=hk['num_head_channels'], num_conv_layers=hk['num_conv_layers'],
                 conv_kernel_size=hk['conv_kernel_size'], gn_groups=hk['norm_cfg']['num_groups'] if hk['norm_cfg'] else None,
                 share_head_flag=hk['share_head_flag'], merge_path_flag=hk['merge_path_flag'],
                 classification_loss_type=spec['classification_loss_type'], regression_loss_type=spec['regression_loss_type'],
                 regression_ranges=spec['regression_ranges'], distance_to_bbox_mode=spec['distance_to_bbox_mode'])
    return a


def oracle_forward(name, x):
    spec = configs.SIBLINGS[name]
    model = configs.build_sibling_model(name, seed=1)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    if spec['meta'] == 'FCOS':
        hk = spec['head']
        head = dict(num_layers=hk['num_layers'], norm=('GroupNorm', hk['norm_cfg']['num_groups']) if hk['norm_cfg'] else None)
        return sibling_oracle.fcos_forward(sd, oracle_arch(spec), oracle_neck(spec), head, x), model
    v1 = None
    if spec['head']['kind'] == 'LFDHeadV1':
        hk = spec['head']
        nc = hk['norm_cfg']
        v1 = dict(num_conv_layers=hk['num_conv_layers'], merge_path_flag=hk['merge_path_flag'],
                  norm=None if nc is None else ('BatchNorm2d' if nc['type'] == 'BatchNorm2d' else ('GroupNorm', nc['num_groups'])),
                  union=spec['regression_loss_type'] in ('IoULoss', 'GIoULoss', 'DIoULoss', 'CIoULoss'))
    cls, reg, sizes = sibling_oracle.lfdv2_forward(sd, oracle_arch(spec), oracle_neck(spec), x, head_v1=v1)
    return (cls, reg, None, sizes), model


def model_input(g):
    n, h, w = [int(v) for v in g['shape']]
    return torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(int(g['x_seed']))) * 2 - 1


def results_args(name_or_case):
    """(ce, decode, strides, ranges, pre, post) for oracle get_results_single"""
    if isinstance(name_or_case, str):
        spec = configs.SIBLINGS[name_or_case]
        if spec['meta'] == 'FCOS':
            return False, 'distance', None, spec['regress_ranges'], spec['pre_nms_bbox_limit'], spec['post_nms_bbox_limit']
        ce = spec['classification_loss_type'] == 'CrossEntropyLoss'
        indep = spec['regression_loss_type'] in ('SmoothL1Loss', 'MSELoss')
        return ce, 'independent' if indep else spec['distance_to_bbox_mode'], None, spec['regression_ranges'], \
            spec['pre_nms_bbox_limit'], spec['post_nms_bbox_limit']
    c = name_or_case
    if c['meta'] == 'FCOS':
        return False, 'distance', c['strides'], [(0, 1)] * len(c['sizes']), c['pre'], c['post']
    return c['ce'], 'independent' if c['loss'] == 'SmoothL1Loss' else c['mode'], c['strides'], c['ranges'], c['pre'], c['post']


def assert_results_equal(got_rows, ref_rows, tol=1e-5):
    assert len(got_rows) == len(ref_rows)
    for a, b in zip(got_rows, ref_rows):
        assert int(a[0]) == int(b[0])
        np.testing.assert_allclose(a[1:], b[1:], rtol=tol, atol=tol)


@pytest.mark.parametrize('name', NAMES)
def test_host_mirror_state_dict_equals_the_reference(name):
    """same keys, shapes and (synthetic, per-key) values as the reference modules: sha of the sorted state_dict"""
    g = load_golden('ref_sibling_%s.npz' % name)
    assert _sha(configs.build_sibling_model(name, seed=1).state_dict()) == str(g['sha'])


@pytest.mark.parametrize('name', NAMES)
def test_sibling_oracle_forward_vs_reference(name):
    g = load_golden('ref_sibling_%s.npz' % name)
    (cls, reg, ctr, sizes), _ = oracle_forward(name, model_input(g))
    assert [list(s) for s in sizes] == g['sizes'].tolist()
    np.testing.assert_allclose(cls.numpy(), g['cls'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(reg.numpy(), g['reg'], rtol=1e-4, atol=2e-5)
    if ctr is not None:
        np.testing.assert_allclose(ctr.numpy(), g['ctr'], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', SC.NECK_CASES, ids=[c[0] for c in SC.NECK_CASES])
def test_pyramid_neck_oracle_vs_reference(case):
    name, kind, kw = case
    g = load_golden('ref_sibling_necks.npz')
    from lfd_amd.model import neck as N
    neck = getattr(N, kind)(num_input_channels_list=[64, 64, 128], num_input_strides_list=[8, 16, 32], **kw)
    configs.synthetic_weights(neck, seed=3)
    assert _sha(neck.state_dict()) == str(g['%s_sha' % name])
    assert list(neck.num_output_strides_list) == g['%s_strides' % name].tolist()
    norm = None
    if kw.get('norm_on_lateral'):
        c = kw['norm_cfg']
        norm = 'BatchNorm2d' if c['type'] == 'BatchNorm2d' else ('GroupNorm', c['num_groups'])
    od = dict(kind=kind, num_inputs=3, num_outputs=kw['num_outputs'], extra_on_input=kw.get('extra_on_input', False),
              extra_type=kw.get('extra_type', 'conv'), norm_on_lateral=norm, relu_on_lateral=kw.get('relu_on_lateral', False),
              relu_before_extra=kw.get('relu_before_extra', kind == 'SimpleFPN'), neighbouring_mode=kw.get('neighbouring_mode', False))
    sd = {k: v.detach() for k, v in neck.state_dict().items()}
    outs = sibling_oracle.pyramid_neck_forward(sd, od, SC.neck_inputs(name), pfx='')
    assert len(outs) == kw['num_outputs']
    for i, y in enumerate(outs):
        np.testing.assert_allclose(y.numpy(), g['%s_out%d' % (name, i)], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', NAMES)
def test_sibling_oracle_get_results_on_model_outputs_vs_reference(name):
    g = load_golden('ref_sibling_%s.npz' % name)
    spec = configs.SIBLINGS[name]
    ce, decode, _, ranges, pre, post = results_args(name)
    n, H, W = [int(v) for v in g['shape']]
    sizes = [tuple(s) for s in g['sizes'].tolist()]
    strides = list(configs.build_sibling_model(name)._point_strides)
    ctr = g['ctr'] if spec['meta'] == 'FCOS' else None
    for key, (hh, ww, sc) in (('results', (H, W, 1.0)), ('results_scaled', (H - 6, W - 10, 0.5))):
        ref = json.loads(str(g[key]))
        for i in range(n):
            dets, labels, _ = sibling_oracle.get_results_single(g['cls'][i], g['reg'][i], None if ctr is None else ctr[i], sizes,
                                                                strides, ranges, ce, decode, float(g['results_thr']),
                                                                float(g['results_iou']), pre, post, (hh, ww), sc)
            assert_results_equal(net_oracle.pack_results(dets, labels), ref[i])


@pytest.mark.parametrize('ci', range(len(SC.RESULT_CASES)))
def test_sibling_oracle_get_results_on_seeded_predictions_vs_reference(ci):
    """pre-NMS top-k on levels of up to 6144 points, centerness factor, post-NMS cap, all decode modes"""
    case = SC.RESULT_CASES[ci]
    g = load_golden('ref_sibling_results.npz')
    ref = json.loads(str(g['results_%d' % ci]))
    cls, reg, ctr = SC.result_inputs(case)
    ce, decode, strides, ranges, pre, post = results_args(case)
    Hh, Ww = case['sizes'][0][0] * case['strides'][0], case['sizes'][0][1] * case['strides'][0]
    metas = [(Hh, Ww, 1.0), (Hh - 9, Ww - 14, 0.75)]
    for i, (hh, ww, sc) in enumerate(metas):
        dets, labels, _ = sibling_oracle.get_results_single(cls[i], reg[i], None if ctr is None else ctr[i], case['sizes'], strides,
                                                            ranges, ce, decode, float(g['thr_%d' % ci]), case['iou'], pre, post,
                                                            (hh, ww), sc)
        assert_results_equal(net_oracle.pack_results(dets, labels), ref[i])


@pytest.mark.parametrize('name', NAMES)
def test_host_target_assignment_vs_reference(name):
    """FCOS.annotation_to_target (fcos.py:108-209) / LFDv2.annotation_to_target (lfdv2.py:232-418): host tensor algebra, as in
    the reference -- exact"""
    g = load_golden('ref_sibling_%s.npz' % name)
    spec = configs.SIBLINGS[name]
    model = configs.build_sibling_model(name, seed=1)
    n, H, W = [int(v) for v in g['shape']]
    for i, hw in enumerate(g['sizes'].tolist()):
        model._head_indexes_to_feature_map_sizes[i] = tuple(hw)
    ann = SC.synth_annotations(5, n, H, W, spec['head']['num_classes'])
    pts = model.generate_point_coordinates(model.head_indexes_to_feature_map_sizes)
    ct, rt = model.annotation_to_target(pts, [torch.from_numpy(b) for b, _ in ann], [torch.from_numpy(l) for _, l in ann])
    np.testing.assert_array_equal(ct.numpy(), g['cls_target'])
    np.testing.assert_array_equal(rt.numpy(), g['reg_target'])
    assert (g['cls_target'] != (spec['head']['num_classes'] if spec['meta'] == 'FCOS' else 0)).any()   # some positives


def _fcosv1_model_and_annotations():
    g = load_golden('ref_sibling_FCOS_FPN.npz')
    spec = dict(configs.SIBLINGS['FCOS_FPN'], meta='FCOSv1')
    from lfd_amd.model import backbone as B, head as H, losses as L, neck as N
    from lfd_amd import model as M
    model = configs.build_sibling(spec, B, N, H, M, L, seed=1)
    n, H_, W_ = [int(v) for v in g['shape']]
    for i, hw in enumerate(g['sizes'].tolist()):
        model._head_indexes_to_feature_map_sizes[i] = tuple(hw)
    return g, model, SC.synth_annotations_overlapping(5, n, H_, W_, spec['head']['num_classes'])


def test_fcosv1_multi_label_targets_vs_reference():
    """fcos.py:550-656: per-class binary targets (0 = present), several classes per point where boxes nest -- exact"""
    from lfd_amd.model import FCOS, FCOSv1
    _, model, ann = _fcosv1_model_and_annotations()
    assert isinstance(model, FCOSv1) and isinstance(model, FCOS)
    ref = load_golden('ref_sibling_FCOSV1.npz')
    pts = model.generate_point_coordinates(model.head_indexes_to_feature_map_sizes)
    ct, rt = model.annotation_to_target(pts, [torch.from_numpy(b) for b, _ in ann], [torch.from_numpy(l) for _, l in ann])
    np.testing.assert_array_equal(ct.numpy(), ref['cls_target'])
    np.testing.assert_array_equal(rt.numpy(), ref['reg_target'])
    assert ((ref['cls_target'] == 0).sum(-1) > 1).sum() > 10       # the fixture does contain multi-label points


def test_fcos_param_groups_follow_the_reference_rule():
    """fcos.py:53-80: conv biases (not norm biases) form the second group with its own lr / weight decay"""
    from lfd_amd.model import FCOS
    m = configs.build_sibling_model('FCOS_FPN')
    assert not isinstance(m.get_param_groups_for_optimizer(), list)
    m._param_groups_cfg = dict(bias_lr=0.02, bias_weight_decay=0.0)
    groups = m.get_param_groups_for_optimizer()
    assert groups[1]['lr'] == 0.02 and groups[1]['weight_decay'] == 0.0
    nb = sum(1 for k, mod in m.named_modules() if not isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.GroupNorm))
             for pn, _ in mod.named_parameters(recurse=False) if 'bias' in pn)
    assert len(groups[1]['params']) == nb > 0
    assert len(groups[0]['params']) + nb == len(list(m.parameters()))
    assert isinstance(m, FCOS)


def test_sibling_modules_refuse_cpu_tensors():
    """no CPU fallback: the product path fails loudly without the device"""
    m = configs.build_sibling_model('FCOS_FPN').eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))
    m2 = configs.build_sibling_model('LFDV2_SFPN').eval()
    with pytest.raises(RuntimeError):
        m2(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        m._neck([torch.zeros(1, 64, 8, 8), torch.zeros(1, 64, 4, 4), torch.zeros(1, 128, 2, 2)])


def test_sibling_oracle_edge_cases():
    """empty result, limits that do not bind, the tie rule of the pre-NMS top-k (lowest point index), cap after NMS"""
    sizes, strides, ranges = [(6, 8), (3, 4)], [8, 16], [(4, 64), (64, 256)]
    P = 48 + 12
    g = torch.Generator().manual_seed(2)
    cls = torch.randn(P, 2, generator=g)
    reg = torch.randn(P, 4, generator=g)
    args = dict(sizes=sizes, strides=strides, ranges=ranges, ce_loss=False, decode='exp', iou_thr=0.5, clamp_hw=(48, 64))
    # nothing above the threshold: the empty-candidate return of multiclass_nms ((0, 4) boxes, nms.py:207-212)
    d, l, p = sibling_oracle.get_results_single(cls, reg, None, score_thr=0.9999, pre_nms_limit=10, post_nms_limit=5, **args)
    assert d.shape[0] == 0 and len(l) == 0 and len(p) == 0
    # a limit >= the level size selects nothing away; post_nms_limit <= 0 caps nothing
    a = sibling_oracle.get_results_single(cls, reg, None, score_thr=0.3, pre_nms_limit=48, post_nms_limit=-1, **args)
    b = sibling_oracle.get_results_single(cls, reg, None, score_thr=0.3, pre_nms_limit=-1, post_nms_limit=-1, **args)
    np.testing.assert_array_equal(a[2], b[2])
    c = sibling_oracle.get_results_single(cls, reg, None, score_thr=0.3, pre_nms_limit=-1, post_nms_limit=3, **args)
    assert len(c[1]) == 3 and np.array_equal(c[2], b[2][:3]) and np.array_equal(c[0], b[0][:3])       # score-descending prefix
    # ties: 20 points share the best key, the limit takes 5 of them -> the five lowest indices (+ the clear winner)
    cls2 = torch.full((P, 1), -3.0)
    cls2[10:30, 0] = 1.0
    cls2[40, 0] = 2.0
    far = torch.zeros(P, 4) - 5.0          # tiny boxes: NMS keeps everything
    d, l, p = sibling_oracle.get_results_single(cls2, far, None, score_thr=0.5, pre_nms_limit=6, post_nms_limit=-1, **args)
    assert sorted(p.tolist()) == [10, 11, 12, 13, 14, 40]
    # centerness factors scale the scores before the threshold (nms.py:192-193, 204)
    ctr = torch.full((P, 1), -20.0)        # sigmoid ~ 2e-9: every product falls below any positive threshold
    d, l, p = sibling_oracle.get_results_single(cls, reg, ctr, score_thr=1e-6, pre_nms_limit=-1, post_nms_limit=-1, **args)
    assert len(l) == 0
