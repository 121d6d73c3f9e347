"""Host-side mirror of the reference interface: constructors, parameter names, properties,
weight folding / packing, plan construction (all on CPU tensors -- no kernel launches)."""
import numpy as np
import pytest
import os
import torch
import torch.nn as nn

from lfd_amd import configs, engine, ops
from lfd_amd.model import LFD, LFDHead, LFDResNet, SimpleNeck
from lfd_amd.model.backbone import FastBlock, FasterBlock, FastestBlock


def test_state_dict_key_names_match_reference_convention():
    m = configs.build_model('WIDERFACE_LFD_S')
    keys = list(m.state_dict().keys())
    for k in ('_backbone._stem.0.weight', '_backbone._stem.1.running_mean', '_backbone.stage0.0._conv1.weight',
              '_backbone.stage0.0._downsample.0.weight', '_backbone.stage3.2._norm2.bias', '_neck.neck0.0.weight',
              '_neck.neck4.1.running_var', '_head.head0_merge_path.0.weight', '_head.head0_merge_path.1.weight',
              '_head.head0_classification_path.0.bias', '_head.head4_regression_path.0.weight',
              '_head._scales.0._scale'):
        assert k in keys, k
    # shared head: duplicated keys alias the same storage (55 keys / 15 unique params)
    hk = [k for k in keys if k.startswith('_head.')]
    assert len(hk) == 55
    assert len({m.state_dict()[k].data_ptr() for k in hk}) == 15
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 1565386                       # WF-S parameter count


def test_strides_and_channels_properties():
    m = configs.build_model('WIDERFACE_LFD_S')
    assert m._backbone.num_output_strides_list == [8, 16, 32, 64, 64]
    assert m._backbone.num_output_channels_list == [64, 64, 64, 128, 128]
    assert m._neck.num_output_strides_list == [8, 16, 32, 64, 64]
    t = configs.build_model('TT100K_LFD_L')
    assert t._backbone.num_output_strides_list == [4, 8, 16, 32]
    assert t._head.num_cls_channels == 46
    assert sum(p.numel() for p in t.parameters()) == 1862646
    assert sum(p.numel() for p in configs.build_model('WIDERFACE_LFD_XS').parameters()) == 898794


def test_state_dict_roundtrip_strict():
    a = configs.build_model('WIDERFACE_LFD_XS', seed=1)
    b = configs.build_model('WIDERFACE_LFD_XS', seed=2)
    b.load_state_dict(a.state_dict(), strict=True)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k


def test_block_variants_register_like_the_reference():
    for cls, n in ((FastBlock, 3), (FasterBlock, 2), (FastestBlock, 2)):
        ds = nn.Sequential(nn.Conv2d(32, 64, 1, 2, bias=False), nn.BatchNorm2d(64))
        blk = cls(32, 64, stride=2, downsample=ds, norm_cfg=dict(type='BatchNorm2d'))
        names = [n_ for n_, _ in blk.named_children()]
        assert names[0] == '_downsample' and names[1:4] == ['_conv1', '_norm1', '_activation']
        assert blk.num_convs == n
    with pytest.raises(AssertionError):
        FasterBlock(32, 64, stride=1, downsample=nn.Sequential())


def test_train_mode_freeze_semantics():
    bb = LFDResNet(block_mode='faster', stem_mode='faster', body_mode=None, stem_channels=32,
                   body_architecture=[1, 1], body_channels=[32, 32], out_indices=((0, 0), (1, 0)), frozen_stages=1,
                   norm_eval=True)
    bb.train()
    assert not bb._stem.training and not bb.stage0[0].training and bb.stage1[0].training
    assert all(not p.requires_grad for p in bb._stem.parameters())
    assert all(not m.training for m in bb.modules() if isinstance(m, nn.BatchNorm2d))


def test_bn_fold_matches_eval_batchnorm():
    torch.manual_seed(0)
    conv = nn.Conv2d(8, 16, 3, padding=1, bias=False)
    bn = nn.BatchNorm2d(16)
    bn.running_mean.normal_(0, .3)
    bn.running_var.uniform_(.5, 1.5)
    bn.weight.data.uniform_(.5, 1.5)
    bn.bias.data.normal_(0, .2)
    bn.eval()
    w, b = engine.fold_conv_norm(conv, bn)
    x = torch.randn(2, 8, 9, 11)
    ref = bn(conv(x))
    # channel widths the MFMA kernels are not instantiated for come back zero-padded to 32 / 64 / 128 (engine.pad_channels):
    # the real channels are unchanged, the padded outputs are exactly zero, padded inputs meet zero weights
    assert tuple(w.shape) == (32, 32, 3, 3) and tuple(b.shape) == (32,)
    assert float(w[16:].abs().max()) == 0.0 and float(w[:, 8:].abs().max()) == 0.0 and float(b[16:].abs().max()) == 0.0
    xp = torch.cat([x, torch.randn(2, 24, 9, 11)], 1)
    got = torch.nn.functional.conv2d(xp, w, b, padding=1)
    torch.testing.assert_close(got[:, :16], ref, atol=2e-6, rtol=1e-5)
    assert float(got[:, 16:].abs().max()) == 0.0
    assert [engine.pad_channels(c) for c in (3, 32, 48, 64, 96, 128)] == [3, 32, 64, 64, 128, 128]


def test_pack_conv_weight_fragment_order():
    w = torch.arange(64 * 32 * 9, dtype=torch.float32).reshape(64, 32, 3, 3) % 1000   # fp16-exact integers
    p = ops.pack_conv_weight(w)
    assert p.shape == (2, 18, 64, 8) and p.dtype == torch.float16
    for tile, k, lane, j in ((0, 0, 0, 0), (1, 7, 45, 3), (0, 17, 63, 7), (1, 9, 31, 5)):
        tap, q = divmod(k, 2)
        r, s = divmod(tap, 3)
        co = tile * 32 + (lane & 31)
        ci = 16 * q + 8 * (lane >> 5) + j
        assert float(p[tile, k, lane, j]) == float(w[co, ci, r, s])


def test_pack_stem_weight_slot_order():
    w = torch.arange(32 * 27, dtype=torch.float32).reshape(32, 3, 3, 3)
    p = engine.pack_stem_weight(w)
    assert p.shape == (1, 2, 64, 8)
    e = lambda co, r, s, c: float(w[co, c, r, s])       # noqa: E731
    assert float(p[0, 0, 5, 4]) == e(5, 0, 1, 1)         # step0 half0: row0, e=4 -> s=1,c=1
    assert float(p[0, 0, 32 + 5, 7]) == e(5, 1, 2, 1)    # step0 half1: row1, e=7 -> s=2,c=1
    assert float(p[0, 1, 9, 0]) == e(9, 2, 0, 0)         # step1 half0: row2
    assert float(p[0, 1, 32 + 9, 1]) == e(9, 1, 2, 2)    # step1 half1: (row1, e=8)
    assert float(p[0, 1, 32 + 9, 3]) == 0.0


def test_engine_plan_structure_on_cpu():
    m = configs.build_model('WIDERFACE_LFD_S')
    m.eval()
    plan = engine.EnginePlan(m._backbone, m._neck, m._head, torch.device('cpu'))
    # the whole 'faster' stem is one kernel for NHWC fp16 frames (stem_fused); for other input formats stem pair 1
    # runs in the stem kernel and stem pair 2 (stem_second) as a conv with a chained 1x1 tail
    assert plan.stem_first[0] == 64 and plan.stem_first[3] is not None
    assert plan.stem_fused is not None and plan.stem_second is not None
    assert plan.stem_second.tail is not None and plan.stem_second.stride == 2 and plan.stem_second.ks == 3
    n3 = sum(2 if c.blk is not None else 1 for c in plan.convs if c.ks == 3)
    n1 = sum(1 for c in plan.convs if c.ks == 1)
    assert (n3, n1) == (2 * 11, 0)              # 11 FasterBlocks; the 4 downsample branches ride on conv1
    # the 64-channel blocks without a downsample branch are ONE launch each (csrc/block.hip): 3 + 1 + 1 of them
    assert sum(1 for c in plan.convs if c.blk is not None) == 5 and len(plan.convs) == 22 - 5
    assert all(c.res == c.src and c.cin == c.cout == 64 for c in plan.convs if c.blk is not None)
    assert sum(1 for c in plan.convs if c.ds is not None) == 4
    # the three 64 -> 64 first blocks of a stage can run as ONE launch (csrc/down.hip): the stride-2 conv points at the conv that
    # closes its block; the 64 -> 128 one cannot.  run_backbone decides per shape (engine._use_fused_down)
    marked = [i for i, c in enumerate(plan.convs) if c.down is not None]
    assert len(marked) == 3 and all(plan.convs[i].down == i + 1 and plan.convs[i].ds is not None and plan.convs[i].cout == 64
                                     and plan.convs[i + 1].res == plan.convs[i].ds[2] and plan.convs[i + 1].ks == 3 for i in marked)
    assert engine._use_fused_down(8, 270, 480) and engine._use_fused_down(8, 68, 120) and engine._use_fused_down(1, 540, 960)
    assert not engine._use_fused_down(1, 270, 480) and not engine._use_fused_down(8, 34, 60)
    # the two 128-channel blocks without branch of the last stage can run as ONE launch each on small maps (csrc/block128.hip)
    m128 = [i for i, c in enumerate(plan.convs) if c.blk128 is not None]
    assert len(m128) == 2 and all(plan.convs[i].blk128 == i + 1 and plan.convs[i + 1].res == plan.convs[i].src
                                   and plan.convs[i + 1].src == plan.convs[i].dst and plan.convs[i].cin == 128 for i in m128)
    assert engine._use_fused_block128(8, 17, 30) and engine._use_fused_block128(1, 34, 60) and not engine._use_fused_block128(64, 17, 30)
    assert len(plan.taps) == 5 and len(plan.levels) == 5
    assert [lv.cin for lv in plan.levels] == [64, 64, 64, 128, 128]
    assert all(len(lv.towers) == 1 and lv.towers[0].reg_rows == 4 and lv.towers[0].cls_rows == 1 for lv in plan.levels)
    t = configs.build_model('TT100K_LFD_L')
    t.eval()
    plan = engine.EnginePlan(t._backbone, t._neck, t._head, torch.device('cpu'))
    assert all(len(lv.towers) == 2 for lv in plan.levels)
    assert plan.levels[0].towers[0].cls_rows == 46 and plan.levels[0].towers[1].reg_rows == 4


def test_unsupported_configs_fail_loudly():
    bb = LFDResNet(block_mode='faster', stem_mode='fast', body_mode=None, stem_channels=64, body_architecture=[1],
                   body_channels=[64], out_indices=((0, 0),), norm_cfg=dict(type='GroupNorm', num_groups=8))
    with pytest.raises(RuntimeError, match='unsupported configuration'):
        engine.EnginePlan(bb, None, None, torch.device('cpu'))


def test_no_cpu_fallback():
    m = configs.build_model('WIDERFACE_LFD_XS').eval()
    with pytest.raises(RuntimeError, match='MI355X only'):
        m(torch.zeros(1, 3, 64, 64))
    from lfd_amd.model.utils import batched_nms
    with pytest.raises(RuntimeError):          # the hot-path NMS (class offsets on the device) has no CPU implementation
        batched_nms(torch.zeros(3, 4), torch.zeros(3), torch.zeros(3, dtype=torch.long), dict(type='nms', iou_thr=0.5))
    from lfd_amd.model.losses.libs import sigmoid_focal_loss_ext as ext
    with pytest.raises(RuntimeError, match='not implemented on the CPU'):   # reference: sigmoid_focal_loss_ext.cpp:32
        ext.forward(torch.zeros(2, 1), torch.zeros(2, dtype=torch.long), 1, 2.0, 0.25)


def test_point_grid_and_gray_ranges():
    m = configs.build_model('WIDERFACE_LFD_S')
    pts = m.generate_point_coordinates({0: (2, 3), 1: (1, 2), 2: (1, 1), 3: (1, 1), 4: (1, 1)})
    assert pts[0].tolist() == [[0, 0], [8, 0], [16, 0], [0, 8], [8, 8], [16, 8]]     # x fastest, no half-stride offset
    assert pts[1].tolist() == [[0, 0], [16, 0]] and pts[0].dtype == torch.int64
    assert m._gray_ranges == [(3, 22), (18, 44), (36, 88), (72, 176), (144, 352)]


def test_target_assignment_and_loss_match_golden_on_cpu_tensors():
    """annotation_to_target / get_loss glue (tensor algebra) against the reference's outputs; the
    focal/IoU kernels need the GPU, so only targets are checked here."""
    from conftest import load_golden
    g = load_golden('ref_model_WIDERFACE_LFD_XS.npz')
    m = configs.build_model('WIDERFACE_LFD_XS')
    for i, s in enumerate(g['sizes'].tolist()):
        m._head_indexes_to_feature_map_sizes[i] = tuple(s)
    pts = m.generate_point_coordinates(m.head_indexes_to_feature_map_sizes)
    cnt, o, bb, ll = g['ann_counts'].tolist(), 0, [], []
    for c in cnt:
        bb.append(torch.from_numpy(g['ann_boxes'][o:o + c]))
        ll.append(torch.from_numpy(g['ann_labels'][o:o + c]))
        o += c
    ct, rt = m.annotation_to_target(pts, bb, ll)
    np.testing.assert_array_equal(ct.numpy(), g['cls_targets'])
    np.testing.assert_array_equal(rt.numpy(), g['reg_targets'])


def test_flat_sgd_layout_and_loud_cpu_failure():
    """lfd_amd.optim.SGD: parameters / gradients / momentum become views of one contiguous buffer per group (values and
    state_dict unchanged, torch.optim.SGD's param_groups layout); the update itself is a HIP kernel and refuses CPU."""
    from lfd_amd import optim
    m = configs.build_model('WIDERFACE_LFD_XS')
    before = {k: v.clone() for k, v in m.state_dict().items()}
    o = optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    ref = torch.optim.SGD(configs.build_model('WIDERFACE_LFD_XS').parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    assert all(torch.equal(before[k], v) for k, v in m.state_dict().items())
    assert {k: v for k, v in o.state_dict()['param_groups'][0].items() if k in o.defaults} == \
        {k: v for k, v in ref.state_dict()['param_groups'][0].items() if k in o.defaults}
    assert o.state_dict()['param_groups'][0]['params'] == ref.state_dict()['param_groups'][0]['params']
    assert o.state_dict()['state'] == {}
    fg = o._flat[0]
    assert fg.numel >= sum(p.numel() for p in m.parameters()) and fg.numel % 64 == 0
    for p, off in zip(fg.params, fg.offsets):
        assert off % 64 == 0 and p.data_ptr() == fg.p.data_ptr() + 4 * off and p.grad.data_ptr() == fg.g.data_ptr() + 4 * off
    m.train()
    c, r = m(torch.randn(2, 3, 128, 128))
    (c.sum() + r.sum()).backward()                    # autograd accumulates into the flat views in place
    assert float(fg.g.abs().sum()) > 0 and fg.adopt_grads()
    assert all(p.grad.data_ptr() == fg.g.data_ptr() + 4 * off for p, off in zip(fg.params, fg.offsets))
    o.zero_grad()
    assert float(fg.g.abs().sum()) == 0 and all(p.grad is not None for p in fg.params)
    with pytest.raises(RuntimeError, match='no CPU path'):
        o.step()
    with pytest.raises(RuntimeError, match='no CPU path'):
        optim.clip_grad_norm_(m.parameters(), 10)
    with pytest.raises(ValueError):
        optim.SGD(m.parameters(), lr=-1)
    with pytest.raises(ValueError):
        optim.SGD(m.parameters(), lr=0.1, nesterov=True)


def test_optimizer_hook_mirror_reads_the_reference_config_keys():
    """lfd_amd.train.OptimizerHook: same ctor as optimizer_hook.py:10-19 ('duration' popped, default = all epochs)
    and the after_train_iter(executor) contract on executor.config_dict (:26-36), here with a torch optimizer on CPU
    (the generic branch: torch clip_grad_norm_ + step)."""
    from lfd_amd import train
    h = train.OptimizerHook(dict(max_norm=10, norm_type=2, duration=5), training_epochs=100)
    assert h._grad_clip_duration == 5 and h._grad_clip_cfg == dict(max_norm=10, norm_type=2)
    assert train.OptimizerHook(dict(max_norm=1), training_epochs=7)._grad_clip_duration == 7
    assert train.OptimizerHook(None, 3)._grad_clip_cfg is None
    w = nn.Parameter(torch.tensor([3.0, 4.0]))
    opt = torch.optim.SGD([w], lr=0.1)

    class Ex(object):
        config_dict = dict(optimizer=opt, loss=(w * torch.tensor([30.0, 40.0])).sum(), model=None, epoch=0)
    h.after_train_iter(Ex)
    assert float(Ex.config_dict['grad_norm']) == pytest.approx(50.0)
    assert torch.allclose(w.detach(), torch.tensor([3.0 - 0.1 * 6.0, 4.0 - 0.1 * 8.0]), atol=1e-5)   # clipped to norm 10
    Ex.config_dict.update(loss=(w * 2).sum(), epoch=5)                                      # past the duration
    h.after_train_iter(Ex)
    assert Ex.config_dict['grad_norm'] == 0


def test_checkpoint_round_trip_in_the_reference_format(tmp_path):
    """lfd_amd.checkpoint: files in the reference's layout (utils.py:90-122) incl. the DataParallel 'module.' prefix and
    the duplicated keys of the shared head load with strict=True; optimizer / scheduler state ride along."""
    from lfd_amd import checkpoint, optim
    m = configs.build_model('WIDERFACE_LFD_S')
    configs.perturb_weights(m, seed=3)
    sd = m.state_dict()
    assert sd['_head.head0_merge_path.0.weight'].data_ptr() == sd['_head.head3_merge_path.0.weight'].data_ptr()   # aliases
    opt = optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[5, 7], gamma=0.1)
    path = str(tmp_path / 'sub' / 'epoch_1.pth')
    checkpoint.save_checkpoint(m, path, optimizer=opt, lr_scheduler=sched, meta=dict(epoch=1))
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'meta', 'state_dict', 'optimizer_state_dict', 'lr_scheduler_state_dict'}
    assert ck['meta']['epoch'] == 1 and 'time' in ck['meta'] and list(ck['state_dict']) == list(sd)
    # a DataParallel-style file (what the reference's Executor writes, executor.py:39,126-132)
    ck['state_dict'] = {'module.' + k: v for k, v in ck['state_dict'].items()}
    path2 = str(tmp_path / 'dp.pth')
    torch.save(ck, path2)
    fresh = configs.build_model('WIDERFACE_LFD_S', seed=1)
    out = checkpoint.load_checkpoint(fresh, path2, strict=True)
    assert out['meta']['epoch'] == 1
    for (k, a), b in zip(fresh.state_dict().items(), sd.values()):
        assert torch.equal(a, b), k
    with pytest.raises(IOError):
        checkpoint.load_checkpoint(fresh, str(tmp_path / 'nope.pth'))
    torch.save({'weights': 1}, str(tmp_path / 'bad.pth'))
    with pytest.raises(RuntimeError, match='No state_dict'):
        checkpoint.load_checkpoint(fresh, str(tmp_path / 'bad.pth'))
    with pytest.raises(TypeError):
        checkpoint.save_checkpoint(m, path, meta=3)


def test_uint8_normalisation_shortcut_is_exact_for_all_256_inputs():
    """k_stem2x<U8> converts frame bytes with fp16(f * (2/255) - 1) (one multiply + one subtract in fp32) instead of the
    reference's (f/255 - 0.5)/0.5 (simple_normalize, augmentation_pipeline.py:31-36): identical fp16 for every byte value."""
    f = np.arange(256, dtype=np.float32)
    ref = ((f / np.float32(255) - np.float32(0.5)) / np.float32(0.5)).astype(np.float16)
    fast = ((f * (np.float32(2) / np.float32(255))) - np.float32(1)).astype(np.float16)
    assert np.array_equal(ref.view(np.uint16), fast.view(np.uint16))


def test_training_schedule_covers_every_parameter_once():
    """train_engine.build_network: the hand-written training schedule (conv / norm / ReLU units + per-level output convs)
    touches every parameter of the six shipped configurations exactly once (shared head towers deduplicated), taps and
    residual wiring follow lfd_resnet.py:96-154 / :458-468; configurations outside its coverage are refused."""
    from lfd_amd import train_engine as te
    # TrafficLight configs (norm-free head; TL_LFD_S also a 48-channel stem): the documented training fallback -- backbone on
    # the HIP training kernels where its channel counts allow, neck / head through PyTorch-ROCm autograd
    tl = {n: configs.build_model(n).train() for n in ('TL_LFD_L', 'TL_LFD_S')}
    assert not te.network_supported(tl['TL_LFD_L']) and not te.network_supported(tl['TL_LFD_S'])
    assert te.supported(tl['TL_LFD_L']._backbone) and not te.supported(tl['TL_LFD_S']._backbone)
    for name in configs.ARCHS:
        if name.startswith('TL_'):
            continue
        m = configs.build_model(name).train()
        assert te.network_supported(m), name
        units, outs = te.build_network(m)
        ps = te.network_params(units, outs)
        assert len({id(p) for p in ps}) == len(ps)
        assert {id(p) for p in ps} == {id(p) for p in m.parameters()}, name
        acts = {0}
        for u in units:
            assert u.src in acts and (u.res is None or u.res in acts) and u.dst not in acts
            acts.add(u.dst)
        assert all(o.src in acts for o in outs)
        assert sorted({o.level for o in outs}) == list(range(m._num_heads))
    frozen = configs.build_model('WIDERFACE_LFD_XS').train()
    frozen._neck.neck0[0].weight.requires_grad_(False)
    assert not te.network_supported(frozen) and te.supported(frozen._backbone)     # frozen parameter -> autograd path
    frozen._backbone._stem[0].weight.requires_grad_(False)
    assert not te.supported(frozen._backbone)
    m = configs.build_model('WIDERFACE_LFD_S').train()
    units, _ = te.build_network(m)
    blk = [u for u in units if u.conv is m._backbone.stage0[0]._conv2][0]
    ds = [u for u in units if u.conv is m._backbone.stage0[0]._downsample[0]][0]
    assert blk.res == ds.dst and ds.relu is False and blk.relu is True          # identity = norm(conv1x1 s2(x)), no ReLU
    plain = [u for u in units if u.conv is m._backbone.stage0[1]._conv2][0]
    first = [u for u in units if u.conv is m._backbone.stage0[1]._conv1][0]
    assert plain.res == first.src                                               # identity = the block input
    # the first conv of each stem pair never stores its activation: the 1x1 conv that follows normalises its operand itself
    full = te.build_network(m)
    d = te._deferred_units(*full)
    assert d == {0: 1, 2: 3} and all(units[v].conv.kernel_size == (1, 1) and units[u].res is None for u, v in d.items())
    assert te._deferred_units(*te.build_network(configs.build_model('WIDERFACE_LFD_XS').train())) == {}     # 32-channel stem
    m.eval()
    assert not te.network_supported(m)            # BatchNorm in eval mode: running statistics, not this path
    m.train()
    m._backbone._stem[1].eval()
    assert not te.supported(m._backbone)
    m._backbone._stem[1].train()
    m._head._conv_kernel_size = 3
    assert not te.network_supported(m)


def test_output_conv_segments_follow_the_padded_weight_rows(monkeypatch):
    """train_engine._out_segs (what lfd_head_out_split_f16 / lfd_head_out_grad_f16 are told) names exactly the rows
    train_engine._out_weight concatenates: class rows first, then the 4 regression rows, zero rows up to 64; the per-level
    Scale (lfd_head.py:157-185) rides on the regression segment only.  Shared heads give one segment table for every level,
    TT100K's separate towers one single-segment output per kind and level.  LFD_OUT_FUSED=0 is the PyTorch-op A/B path."""
    from lfd_amd import train_engine as te
    for name in ('WIDERFACE_LFD_S', 'TT100K_LFD_L', 'WIDERFACE_LFD_XS'):
        m = configs.build_model(name).train()
        _, outs = te.build_network(m)
        for o in outs:
            wp, bp = te._out_weight(o)
            assert wp.shape[0] == 64 and bp.shape == (64,)
            segs = te._out_segs(o)
            assert [sg['kind'] for sg in segs] == [k for k, _ in o.convs]
            r = 0
            for sg, (kind, conv) in zip(segs, o.convs):
                assert sg['row0'] == r and sg['channels'] == conv.out_channels and sg['conv'] is conv
                assert torch.equal(wp[r:r + conv.out_channels], conv.weight.detach())
                assert torch.equal(bp[r:r + conv.out_channels], conv.bias.detach())
                if kind == 'reg':
                    assert conv.out_channels == 4
                    assert (sg['scale'] is None) == (o.scale is None)
                    if o.scale is not None:
                        assert sg['scale'].data_ptr() == o.scale._scale.data_ptr()
                else:
                    assert sg['scale'] is None and conv.out_channels == m._head.num_cls_channels
                r += conv.out_channels
            assert r <= 64 and not wp[r:].any() and not bp[r:].any()
        kinds = sorted((o.level, k) for o in outs for k, _ in o.convs)
        assert kinds == sorted((l, k) for l in range(m._num_heads) for k in ('cls', 'reg'))     # every level, each kind once
    assert te._fused_outputs()
    monkeypatch.setenv('LFD_OUT_FUSED', '0')
    assert not te._fused_outputs()


def test_every_loss_the_lfd_constructor_accepts_is_provided():
    """lfd.py:52-66: classification loss in {BCEWithLogitsLoss, FocalLoss, CrossEntropyLoss, QualityFocalLoss}, regression
    loss in {SmoothL1Loss, MSELoss} ('independent') or {IoULoss, GIoULoss, DIoULoss, CIoULoss} ('union'): all importable
    from lfd_amd.model with the reference's constructor arguments, and accepted by LFD."""
    import lfd_amd.model as M
    arch = configs.ARCHS['WIDERFACE_LFD_XS']
    base = configs.build_model('WIDERFACE_LFD_XS')
    cls_losses = [M.BCEWithLogitsLoss(reduction='mean', loss_weight=1.0), M.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25),
                  M.CrossEntropyLoss(reduction='mean', loss_weight=1.0), M.QualityFocalLoss(use_sigmoid=True, beta=2.0)]
    reg_losses = [(M.SmoothL1Loss(beta=1.0), 'independent'), (M.MSELoss(), 'independent'), (M.IoULoss(eps=1e-6), 'union'),
                  (M.GIoULoss(eps=1e-6), 'union'), (M.DIoULoss(eps=1e-6), 'union'), (M.CIoULoss(eps=1e-6), 'union')]
    for cl in cls_losses:
        for rl, kind in reg_losses:
            m = M.LFD(backbone=base._backbone, neck=base._neck, head=base._head, num_classes=1,
                      regression_ranges=arch['regression_ranges'], point_strides=base._point_strides,
                      classification_loss_func=cl, regression_loss_func=rl, distance_to_bbox_mode='exp')
            assert m._regression_loss_type == kind
    assert hasattr(M, 'L1Loss')
    with pytest.raises(AssertionError):
        M.QualityFocalLoss(use_sigmoid=False)
    with pytest.raises(AssertionError):
        M.LFD(backbone=base._backbone, neck=base._neck, head=base._head, num_classes=1,
              regression_ranges=arch['regression_ranges'], point_strides=base._point_strides,
              classification_loss_func=M.FocalLoss(), regression_loss_func=M.L1Loss())     # not in LFD's accepted set


def test_archs_match_reference_configs(known_answers):
    """configs.ARCHS is a transcription of the six reference config scripts; the kwargs every constructor receives
    in the reference's prepare_model() were recorded by tests/golden/make_golden_configs.py (executed from the
    reference's own source).  Ctor defaults the configs rely on: LFD.range_assign_mode='dist' (lfd.py:23),
    LFDHead.conv_kernel_size=1 (lfd_head.py:37)."""
    from lfd_amd import configs
    ref = known_answers['reference_model_configs']
    assert sorted(ref) == sorted(configs.ARCHS)
    for name, arch in configs.ARCHS.items():
        r = ref[name]
        bb, neck, head, lfd = r['LFDResNet'], r['SimpleNeck'], r['LFDHead'], r['LFD']
        as_list = lambda v: [list(e) if isinstance(e, (list, tuple)) else e for e in v]
        got = dict(block_mode=bb['block_mode'], stem_mode=bb['stem_mode'], stem_channels=bb['stem_channels'],
                   body_architecture=bb['body_architecture'], body_channels=bb['body_channels'], out_indices=bb['out_indices'],
                   num_neck_channels=neck['num_neck_channels'], num_classes=lfd['num_classes'],
                   num_head_channels=head['num_head_channels'], num_conv_layers=head['num_conv_layers'],
                   conv_kernel_size=head.get('conv_kernel_size', 1), gn_groups=head['norm_cfg']['num_groups'] if head['norm_cfg'] else None,
                   share_head_flag=head['share_head_flag'], merge_path_flag=head['merge_path_flag'],
                   classification_loss_type=head['classification_loss_type'], regression_loss_type=head['regression_loss_type'],
                   regression_ranges=lfd['regression_ranges'], gray_range_factors=lfd['gray_range_factors'],
                   range_assign_mode=lfd.get('range_assign_mode', 'dist'), distance_to_bbox_mode=lfd['distance_to_bbox_mode'])
        assert sorted(got) == sorted(arch), name
        for k, v in got.items():
            mine = arch[k]
            if isinstance(mine, (list, tuple)):
                mine = as_list(mine)
            assert mine == v, (name, k, mine, v)
        # the fixed kwargs configs.build_modules passes
        assert bb['body_mode'] is None and bb['input_channels'] == 3 and bb['frozen_stages'] == -1 and bb['norm_eval'] is False
        assert bb['norm_cfg'] == dict(type='BatchNorm2d') and bb['activation_cfg'] == dict(type='ReLU', inplace=True)
        assert neck['norm_cfg'] == dict(type='BatchNorm2d') and (head['norm_cfg'] is None or head['norm_cfg']['type'] == 'GroupNorm')
        assert head['num_classes'] == lfd['num_classes'] and head['num_heads'] == len(arch['out_indices'])
        assert head['num_input_channels'] == neck['num_neck_channels']
        assert r['IoULoss'] == dict(eps=1e-6, reduction='mean', loss_weight=1.0)
        if arch['classification_loss_type'] == 'FocalLoss':
            assert r['FocalLoss'] == dict(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0)
        elif arch['classification_loss_type'] == 'QualityFocalLoss':
            assert r['QualityFocalLoss'] == dict(use_sigmoid=True, beta=2.0, reduction='mean', loss_weight=2.0)
        else:
            assert r['CrossEntropyLoss'] == dict(reduction='mean', loss_weight=1.0)


def test_flat_sgd_rebinds_parameters_replaced_behind_its_back():
    """The reference builds the optimizer on CPU parameters and moves the model afterwards (executor.py:36-39);
    Module._apply then replaces p.data.  The flat groups must notice and re-bind (here: same device, storage replaced)."""
    from lfd_amd import optim
    m = configs.build_model('WIDERFACE_LFD_XS')
    o = optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    fg = o._flat[0]
    with torch.no_grad():
        for p in m.parameters():
            p.data = p.data.clone() * 2.0          # what .to() / an assign-style load does
    assert any(p.data_ptr() != fg.p.data_ptr() + 4 * off for p, off in zip(fg.params, fg.offsets))
    vals = [p.detach().clone() for p in m.parameters()]
    assert fg.adopt_grads() and fg.rebound
    for p, off, v in zip(fg.params, fg.offsets, vals):
        assert p.data_ptr() == fg.p.data_ptr() + 4 * off and torch.equal(p.detach(), v)
    assert not fg.adopt_params()


def test_extension_modules_are_self_contained_and_serve_cpu_tensors_like_the_reference(tmp_path, known_answers):
    """lfd_amd/model/utils/libs/nms_ext.py and model/losses/libs/sigmoid_focal_loss_ext.py are copied
    into a fresh package tree under the REFERENCE's module names (lfd.model.utils.libs.nms_ext, nms.py:4;
    lfd.model.losses.libs.sigmoid_focal_loss_ext, focal_loss.py:6) and imported in a subprocess that has no lfd_amd on its
    path.  CPU tensors: nms == the reference's compiled nms_cpu (golden ref_nms.npz), soft_nms / nms_match == vectors
    generated with the reference extension (golden ref_nms_cpu_extra.npz); the focal loss refuses CPU tensors like the
    reference (sigmoid_focal_loss_ext.cpp:32,49)."""
    import shutil
    import subprocess
    import sys
    from conftest import GOLDEN, PKG
    from lfd_amd import _lib
    for sub in ('lfd', 'lfd/model', 'lfd/model/utils', 'lfd/model/utils/libs', 'lfd/model/losses', 'lfd/model/losses/libs'):
        os.makedirs(str(tmp_path / sub))
        (tmp_path / sub / '__init__.py').write_text('')
    shutil.copy(os.path.join(PKG, 'lfd_amd', 'model', 'utils', 'libs', 'nms_ext.py'), str(tmp_path / 'lfd/model/utils/libs/nms_ext.py'))
    shutil.copy(os.path.join(PKG, 'lfd_amd', 'model', 'losses', 'libs', 'sigmoid_focal_loss_ext.py'),
                str(tmp_path / 'lfd/model/losses/libs/sigmoid_focal_loss_ext.py'))
    script = tmp_path / 'run.py'
    script.write_text('''
import sys, json, numpy as np, torch
sys.path = [p for p in sys.path if 'lfd-a-light-and-fast-detector_amd' not in p]
sys.path.insert(0, sys.argv[1])
from lfd.model.utils.libs import nms_ext
from lfd.model.losses.libs import sigmoid_focal_loss_ext as fl
assert 'lfd_amd' not in sys.modules
g = np.load(sys.argv[2] + '/ref_nms.npz')
for ci in range(len(g['cases'])):
    keep = nms_ext.nms(torch.from_numpy(g['dets_%d' % ci]), float(g['cases'][ci][1]))
    assert keep.dtype == torch.long and not keep.is_cuda
    np.testing.assert_array_equal(keep.numpy(), g['keep_%d' % ci])
e = np.load(sys.argv[2] + '/ref_nms_cpu_extra.npz')
for ci in range(int(e['num_cases'])):
    d = torch.from_numpy(e['dets_%d' % ci])
    thr, method, sigma, min_score = [float(v) for v in e['params_%d' % ci]]
    out = nms_ext.soft_nms(d, thr, int(method), sigma, min_score)
    np.testing.assert_array_equal(out.numpy(), e['soft_%d' % ci])
    groups = nms_ext.nms_match(d, thr)
    assert [len(x) for x in groups] == e['match_sizes_%d' % ci].tolist()
    assert [i for x in groups for i in x] == e['match_members_%d' % ci].tolist()
assert nms_ext.nms(torch.zeros((0, 5)), 0.5).numel() == 0
for fn, args in ((fl.forward, (torch.zeros(2, 1), torch.zeros(2, dtype=torch.long), 1, 2.0, 0.25)),
                 (fl.backward, (torch.zeros(2, 1), torch.zeros(2, dtype=torch.long), torch.zeros(2, 1), 1, 2.0, 0.25))):
    try:
        fn(*args)
        raise SystemExit('CPU focal loss did not raise')
    except RuntimeError as ex:
        assert 'not implemented on the CPU' in str(ex)
print('ext ok')
''')
    env = dict(os.environ, LFD_HIP_LIB=_lib.LIB_PATH)
    out = subprocess.run([sys.executable, str(script), str(tmp_path), GOLDEN], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'ext ok' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_host_nms_at_baseline_candidate_counts():
    """the product's HOST nms (lfd_nms_cpu_f32 behind lfd_amd.model.utils.nms for CPU tensors, the counterpart of the
    reference's nms_cpu.cpp:7-66) == the compiled reference at K = 4096 / 8192 boxes (ref_nms_large.npz)"""
    from conftest import load_golden
    import nms_large_cases as cases
    from lfd_amd.model.utils import nms
    g = load_golden('ref_nms_large.npz')
    for ci, (k, thr) in enumerate(cases.CASES):
        d = torch.from_numpy(cases.dets(ci))
        sup, inds = nms(d, float(thr))
        assert not inds.is_cuda and inds.dtype == torch.long
        np.testing.assert_array_equal(inds.numpy(), g['keep_%d' % ci], err_msg='case %d' % ci)
        assert torch.equal(sup, d[inds])


def test_host_soft_nms_and_nms_match_on_2000_dense_boxes():
    """nms_ext.soft_nms (hard / linear / gaussian) and nms_ext.nms_match -- CPU-only in the reference (nms_cpu.cpp:76-283) --
    on 2000 densely overlapping boxes == the compiled reference's output rows / groups (ref_nms_large.npz; the fixtures of
    make_golden.py stop at 300 boxes)"""
    from conftest import load_golden
    import nms_large_cases as cases
    from lfd_amd.model.utils.libs import nms_ext
    g = load_golden('ref_nms_large.npz')
    for ci, (k, thr, method, sigma, min_score) in enumerate(cases.SOFT_CASES):
        d = torch.from_numpy(cases.soft_dets(ci))
        out = nms_ext.soft_nms(d, float(thr), int(method), float(sigma), float(min_score))
        np.testing.assert_array_equal(out.numpy(), g['soft_%d' % ci], err_msg='soft case %d' % ci)
        groups = nms_ext.nms_match(d, float(thr))
        assert [len(x) for x in groups] == g['match_sizes_%d' % ci].tolist()
        assert [i for x in groups for i in x] == g['match_members_%d' % ci].tolist()


def test_python_nms_api_on_cpu_tensors_and_numpy(known_answers):
    """lfd_amd.model.utils.nms / soft_nms keep the reference's host behaviour (nms.py:7-116): numpy in -> numpy out, CPU
    tensors stay on the CPU, docstring vectors reproduce."""
    from lfd_amd.model.utils import nms, soft_nms
    ka = known_answers['nms_docstring']
    d = np.array(ka['dets'], np.float32)
    sup, inds = nms(d, ka['iou_thr'])
    assert isinstance(inds, np.ndarray) and inds.tolist() == ka['keep'] and sup.shape == (3, 5)
    sup_t, inds_t = nms(torch.from_numpy(d), ka['iou_thr'])
    assert inds_t.tolist() == ka['keep'] and not inds_t.is_cuda and torch.equal(sup_t, torch.from_numpy(d)[inds_t])
    ks = known_answers['soft_nms_docstring']
    nd, ni = soft_nms(np.array(ks['dets'], np.float32), ks['iou_thr'], sigma=ks['sigma'])
    assert ni.dtype == np.int64 and ni.tolist() == ks['inds'] and len(nd) == ks['expected_len']
    np.testing.assert_allclose(nd, np.array(ks['new_dets'], np.float32), rtol=0, atol=0)
    with pytest.raises(ValueError):
        soft_nms(np.zeros((1, 5), np.float32), 0.5, method='nope')


def _tiny_arch():
    return dict(configs.ARCHS['WIDERFACE_LFD_XS'], body_architecture=[1], body_channels=[64], out_indices=((0, 0),),
                regression_ranges=((4, 320),))


def test_checkpoint_written_by_the_reference_loads_strictly():
    """tests/golden/ref_checkpoint_tiny.pth was written by the REFERENCE's save_checkpoint (execution/utils.py:90-122) from
    the reference's modules wrapped in nn.DataParallel (make_golden_checkpoint.py).  lfd_amd.checkpoint.load_checkpoint
    must take it with strict=True into this package's modules, tensor for tensor, and the optimizer / scheduler states
    must resume in lfd_amd.optim.SGD."""
    from conftest import GOLDEN
    from lfd_amd import checkpoint, optim
    path = os.path.join(GOLDEN, 'ref_checkpoint_tiny.pth')
    m = configs.build_model(_tiny_arch(), seed=123)          # different init: everything must come from the file
    ck = checkpoint.load_checkpoint(m, path, strict=True)
    assert set(ck) == {'meta', 'state_dict', 'optimizer_state_dict', 'lr_scheduler_state_dict'} and ck['meta']['epoch'] == 7
    expect = configs.build_model(_tiny_arch(), seed=666)
    configs.perturb_weights(expect, seed=2)
    assert list(m.state_dict()) == list(ck['state_dict']) == list(expect.state_dict())
    for (k, a), b in zip(m.state_dict().items(), expect.state_dict().values()):
        assert torch.equal(a, b), k
    opt = optim.SGD(m.parameters(), lr=0.5, momentum=0.0)
    opt.load_state_dict(ck['optimizer_state_dict'])
    g = opt.param_groups[0]
    assert (g['lr'], g['momentum'], g['weight_decay']) == (0.1, 0.9, 1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.5)
    sched.load_state_dict(ck['lr_scheduler_state_dict'])
    assert sorted(sched.milestones) == [60, 90] and sched.gamma == 0.1


def test_graphed_train_step_refuses_what_it_cannot_capture():
    """lfd_amd.train.GraphedTrainStep: the flat-buffer SGD and L2 clipping only; a CPU batch is refused at the first call"""
    import torch
    from lfd_amd import configs, optim, train
    m = configs.build_model('WIDERFACE_LFD_XS').train()
    with pytest.raises(RuntimeError):
        train.GraphedTrainStep(m, torch.optim.SGD(m.parameters(), lr=0.1))
    with pytest.raises(RuntimeError):
        train.GraphedTrainStep(m, optim.SGD(m.parameters(), lr=0.1), grad_clip_cfg=dict(max_norm=1.0, norm_type=1))
    step = train.GraphedTrainStep(m, optim.SGD(m.parameters(), lr=0.1), grad_clip_cfg=dict(max_norm=1.0, norm_type=2))
    with pytest.raises(RuntimeError):
        step(torch.zeros(1, 3, 64, 64), [(np.zeros((0, 4), np.float32), np.zeros((0,), np.int64))])


def test_cpu_nms_surface_evaluates_float64_input_in_double_like_the_reference():
    """AT_DISPATCH_FLOATING_TYPES (nms_cpu.cpp:70,212,287): a float64 tensor / numpy array -- numpy's default dtype -- is
    evaluated in double.  tests/golden/ref_nms_cpu_f64.npz holds outputs of the reference's own compiled extension on double
    tensors (make_golden_nms_f64.py), including a case whose float32 evaluation keeps a box the double evaluation
    suppresses."""
    from conftest import GOLDEN
    from lfd_amd.model.utils import nms as nms_py
    from lfd_amd.model.utils.libs import nms_ext
    g = np.load(os.path.join(GOLDEN, 'ref_nms_cpu_f64.npz'))
    nc = int(g['num_cases'])
    for ci in range(nc):
        d = torch.from_numpy(g['dets_%d' % ci])
        assert d.dtype == torch.float64
        thr, method, sigma, min_score = [float(v) for v in g['params_%d' % ci]]
        np.testing.assert_array_equal(nms_ext.nms(d, thr).numpy(), g['keep_%d' % ci])
        soft = nms_ext.soft_nms(d, thr, int(method), sigma, min_score)
        assert soft.dtype == torch.float64
        np.testing.assert_array_equal(soft.numpy(), g['soft_%d' % ci])
        groups = nms_ext.nms_match(d, thr)
        assert [len(x) for x in groups] == g['match_sizes_%d' % ci].tolist()
        assert [i for x in groups for i in x] == g['match_members_%d' % ci].tolist()
        kept, inds = nms_py(g['dets_%d' % ci], thr)            # numpy float64 in -> numpy out (nms.py:36-47)
        assert inds.tolist() == g['keep_%d' % ci].tolist() and kept.dtype == np.float64
    d = torch.from_numpy(g['dets_%d' % (nc - 1)])
    thr = float(g['params_%d' % (nc - 1)][0])
    assert nms_ext.nms(d.float(), thr).tolist() == [0, 1, 2] and nms_ext.nms(d, thr).tolist() == [0, 2]


def test_replay_change_detector_sees_storage_swaps_and_replaced_modules():
    """engine.version_sum guards the captured step graphs (LFD.detect_resident): in-place writes bump version counters;
    `p.data = other`, load_state_dict(assign=True) and a replaced submodule change storage without touching them."""
    from lfd_amd import engine
    m = configs.build_model('WIDERFACE_LFD_XS').eval()
    v0 = engine.version_sum(m)
    assert engine.version_sum(m) == v0
    p = next(m._backbone.parameters())
    with torch.no_grad():
        p.add_(1.0)
    v1 = engine.version_sum(m)
    assert v1 != v0
    p.data = p.data.clone()                       # storage swap, no version bump
    v2 = engine.version_sum(m)
    assert v2 != v1 and v2[:2] == v1[:2]
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)            # parameter OBJECTS replaced: the cached tensor list must be dropped
    assert '_lfd_tensors' not in m.__dict__
    v3 = engine.version_sum(m)
    assert v3 != v2
    other = configs.build_model('WIDERFACE_LFD_XS')
    m._head = other._head                         # replaced submodule
    assert '_lfd_tensors' not in m.__dict__ and engine.version_sum(m) != v3


def test_precision_property_and_unsupported_exception_type():
    from lfd_amd import engine
    m = configs.build_model('WIDERFACE_LFD_XS').eval()
    assert m.precision == 'fp16' and m.PRECISIONS == ('fp16', 'fp32_storage')
    m.precision = 'fp32_storage'
    assert m.precision == 'fp32_storage'
    with pytest.raises(ValueError):
        m.precision = 'bf16'
    with pytest.raises(RuntimeError):             # both modes are HIP-only: a CPU tensor is refused, not emulated
        m(torch.zeros(1, 3, 64, 64))
    assert issubclass(engine.Unsupported, RuntimeError)
    with pytest.raises(engine.Unsupported):
        engine._unsupported('x')


def test_p32_weight_packing_layout():
    """engine_p32.pack_weight: [slab][chunk][tap][kstep][hi | 2^11 lo][lane = 32 khalf + cout][8]; hi + lo / 2^11 restores
    the fp32 weight to ~2^-22 relative"""
    from lfd_amd import engine_p32
    g = torch.Generator().manual_seed(3)
    w = torch.randn(40, 64, 3, 3, generator=g) * torch.logspace(-4, 1, 40)[:, None, None, None]
    pk = engine_p32.pack_weight(w)
    assert pk.shape == (2, 2, 9, 2, 2, 64, 8) and pk.dtype == torch.float16
    rec = pk[:, :, :, :, 0].float() + pk[:, :, :, :, 1].float() / 2048.0     # [slab, chunk, tap, kk, lane, j]
    for (co, ci, dy, dx) in [(0, 0, 0, 0), (39, 63, 2, 2), (33, 17, 1, 2), (7, 40, 0, 1)]:
        slab, col = co // 32, co % 32
        chunk, r = ci // 32, ci % 32
        kk, r = r // 16, r % 16
        khalf, j = r // 8, r % 8
        got = float(rec[slab, chunk, dy * 3 + dx, kk, khalf * 32 + col, j])
        assert abs(got - float(w[co, ci, dy, dx])) <= 2.0 ** -21 * abs(float(w[co, ci, dy, dx])) + 1e-10   # (+ fp16-subnormal floor 2^-25 / 2^11)
    assert float(rec[1, :, :, :, 8:32].abs().max()) == 0 and float(rec[1, :, :, :, 40:64].abs().max()) == 0   # rows 40..63: zero padding


@pytest.mark.parametrize('name', sorted(configs.ARCHS))
def test_precise_plan_covers_every_named_configuration(name):
    """engine_p32.PrecisePlan (host side of LFD.precision = 'fp32_storage'): built on CPU tensors for every named configuration
    -- one conv op per nn.Conv2d of the module tree (a stem pair conv3x3 s2 -> conv1x1 at 64 channels counts as one chained
    launch), one GroupNorm op per tower norm and level, output convs addressed into the [N,P,C'] / [N,P,4] tensors, packed
    weights of the documented size, fp32 biases padded to 32-channel slabs."""
    import torch.nn as nn
    from lfd_amd import engine_p32
    m = configs.build_model(name).eval()
    plan = engine_p32.PrecisePlan(m, torch.device('cpu'))
    convs = [o for o in plan.ops if o.kind == 'conv']
    gns = [o for o in plan.ops if o.kind == 'gn']
    n_conv_modules = sum(isinstance(x, nn.Conv2d) for x in m._backbone.modules())
    head, neck = m._head, m._neck
    nl = head._num_heads
    per_level_head = sum(isinstance(x, nn.Conv2d) for x in
                         list(getattr(head, 'head0_merge_path').modules()) + list(getattr(head, 'head0_classification_path').modules())
                         + list(getattr(head, 'head0_regression_path').modules()))
    expect = n_conv_modules + nl * (1 + per_level_head)
    chained = sum(o.tail is not None for o in convs)
    assert len(convs) + chained == expect, (len(convs), chained, expect)
    stem64 = m._backbone._stem_channels in (48, 64)          # 48 is zero-padded to 64 (TL_LFD_S)
    assert chained == ({'fast': 1, 'faster': 2, 'fastest': 0}[m._backbone._stem_mode] if stem64 else 0)
    has_gn = head._norm_cfg is not None and head._norm_cfg['type'] == 'GroupNorm'
    towers = 1 if head._merge_path_flag else 2
    assert len(gns) == (nl * towers * head._num_conv_layers if has_gn else 0)
    outs = [o for o in convs if o.out is not None]
    assert len(outs) == 2 * nl and {o.out for o in outs} == {'cls', 'reg'} and sorted({o.level for o in outs}) == list(range(nl))
    for o in convs:
        ns = -(-o.cout // 32)
        taps_chunks = 1 if o.patch else (o.cin // 32) * o.ks * o.ks      # the first conv: its 27 taps are ONE 32-wide k chunk
        assert o.w.dtype == torch.float16 and o.w.numel() == ns * taps_chunks * 2 * 2 * 64 * 8
        assert o.b.dtype == torch.float32 and o.b.numel() == ns * 32
        assert o.patch == (o.src == 'input')
    assert len(plan.taps) == nl


def test_planes_weight_packing_and_plane_round_trip():
    """engine_p2 (host side of the hi/lo-plane kernels, csrc/planes.hip): pack_planes_weight = ops.pack_conv_weight order per
    plane, rows zero-padded to 32; hi + lo / 2^11 restores the fp32 weight to ~2^-22; to_planes / from_planes likewise."""
    from lfd_amd import engine_p2, ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(40, 64, 3, 3, generator=g) * torch.logspace(-4, 1, 40)[:, None, None, None]
    pk = engine_p2.pack_planes_weight(w)
    assert pk.shape == (2, 2, 36, 64, 8) and pk.dtype == torch.float16
    wp = torch.cat([w, torch.zeros(24, 64, 3, 3)], 0)
    hi = wp.half()
    assert torch.equal(pk[0], ops.pack_conv_weight(hi)) and torch.equal(pk[1], ops.pack_conv_weight(((wp - hi.float()) * 2048).half()))
    rec = pk[0].float() + pk[1].float() / 2048.0           # [slab, kstep = (tap, q), lane = 32 half + cout, j]
    for (co, ci, dy, dx) in [(0, 0, 0, 0), (39, 63, 2, 2), (33, 17, 1, 2)]:
        got = float(rec[co // 32, (dy * 3 + dx) * 4 + ci // 16, ((ci % 16) // 8) * 32 + co % 32, ci % 8])
        assert abs(got - float(w[co, ci, dy, dx])) <= 2.0 ** -21 * abs(float(w[co, ci, dy, dx])) + 1e-10
    x = torch.randn(2, 5, 7, 32, generator=g) * torch.logspace(-3, 2, 32)
    p = engine_p2.to_planes(x)
    assert p.shape == (2, 2, 5, 7, 32) and p.dtype == torch.float16
    assert float(((engine_p2.from_planes(p) - x).abs() / x.abs().clamp_min(1e-3)).max()) <= 2.0 ** -21
    ws = engine_p2.pack_planes_stem_weight(torch.randn(64, 3, 3, 3, generator=g))
    assert ws.shape == (2, 2, 2, 64, 8)


@pytest.mark.parametrize('name', sorted(configs.ARCHS))
def test_planes_plan_covers_the_shipped_configurations(name):
    """engine_p2.PlanesPlan built on CPU tensors: every WIDERFACE / TT100K configuration gets the plane plan (TL_*: 3x3 head
    towers -> `Unsupported` -> the fp32-tensor plan of engine_p32); every conv it schedules has an instance in lfd_pl_conv2d's
    dispatch (engine_p2._DISPATCH mirrors csrc/planes.hip); three launches per pyramid level with merged towers, the consumer
    of every tower conv carries its GroupNorm; launch ranges for the optional side-stream fork."""
    from lfd_amd import engine_p2, engine_p32
    m = configs.build_model(name).eval()
    if name.startswith('TL_'):
        with pytest.raises(engine_p2.Unsupported):
            engine_p2.PlanesPlan(m, torch.device('cpu'))
        assert isinstance(engine_p32.get_plan(m, torch.device('cpu')), engine_p32.PrecisePlan)
        return
    plan = engine_p2.PlanesPlan(m, torch.device('cpu'))
    assert isinstance(engine_p32.get_plan(m, torch.device('cpu')), engine_p2.PlanesPlan)
    head = m._head
    nl = head._num_heads
    convs = [o for o in plan.ops if o.kind == 'conv']
    assert plan.ops[0].kind == 'stem' and len(plan.level_ops) == nl and len(plan.tap_ready) == nl
    for o in convs:
        opts = engine_p2._DISPATCH[(o.cin, o.ks, o.stride, -(-o.cout // 32))]
        used = {k for k, v in (('tail', o.tail), ('ds', o.ds), ('res', o.res), ('out32', o.out_mode == 2)) if v not in (None, False)}
        assert used <= opts and (o.gn is None or 'gn' in opts)
        assert o.w.shape[0] == 2 and o.w.dtype == torch.float16 and o.b.numel() % 128 == 0
    towers = 1 if head._merge_path_flag else 2
    gn_convs = [o for o in convs if o.gn is not None]
    assert len(gn_convs) == plan.num_gn == nl * towers * head._num_conv_layers
    assert sum(o.gnin is not None for o in convs) == plan.num_gn          # every set of sums has exactly one consumer
    outs = [o for o in convs if o.out_mode == 2]
    assert len(outs) == nl * towers and sorted({o.level for o in outs}) == list(range(nl))
    for a, b in plan.level_ops:
        assert b - a == (3 if head._merge_path_flag else 1 + 2 * head._num_conv_layers + 2)
    # the stage entries carry their identity branch, the closing conv of every block its residual
    assert sum(o.ds is not None for o in convs) == sum(1 for i in range(len(m._backbone._body_architecture)))
    assert sum(o.res is not None for o in convs) == sum(m._backbone._body_architecture)


def test_planes_plan_groups_the_head_by_level_and_fuses_the_faster_stem():
    """engine_p2: one lfd_pl_conv2d_levels launch per head conv over all pyramid levels (split where the neck's input width
    differs), producers of GroupNorm sums in earlier launches than their consumers; the 4-conv 64-channel stem gets the
    one-launch form (lfd_pl_stem2x), the XS model (32 channels) keeps two launches."""
    from lfd_amd import engine_p2
    m = configs.build_model('WIDERFACE_LFD_S').eval()
    plan = engine_p2.PlanesPlan(m, torch.device('cpu'))
    nl = len(plan.level_ops)
    assert plan.stem2x is not None and plan.stem2x.tail is plan.ops[1] and plan.ops[1].tail is not None
    assert tuple(plan.stem2x.w1.shape) == (2, 2, 2, 64, 8) and tuple(plan.stem2x.w2.shape) == (2, 2, 4, 64, 8)
    groups = plan.level_groups
    assert groups is not None and sorted(i for g in groups for i in g) == list(range(plan.head_start, len(plan.ops)))
    assert len(groups) < len(plan.ops) - plan.head_start and max(len(g) for g in groups) == nl
    seen = set()
    for g in groups:
        sig = {(plan.ops[i].cin, plan.ops[i].cout, plan.ops[i].out_mode, plan.ops[i].tail is not None) for i in g}
        assert len(sig) == 1                                       # one kernel instance per launch
        for i in g:
            if plan.ops[i].gnin is not None:
                assert plan.ops[i].gnin[0] in seen                 # its producer ran in an earlier launch
        seen |= {plan.ops[i].gn for i in g if plan.ops[i].gn is not None}
    xs = engine_p2.PlanesPlan(configs.build_model('WIDERFACE_LFD_XS').eval(), torch.device('cpu'))
    assert xs.stem2x is None and xs.level_groups is not None


def test_fused_stem_weight_packings_reproduce_the_convs_through_the_kernel_s_k_order():
    """pack_planes_stem2x_weight / pack_planes_stem2x_tail_weight (lfd_pl_stem2x): emulate what the kernel feeds the matrix
    core -- conv0's B fragment gathered as aligned dwords of the patch rows (a junk half in front, the constant-one slot), the
    32x32 accumulator layout re-used as the 1x1's B fragment -- and contract it with the packed A fragments: == the convs."""
    from lfd_amd import engine_p2
    g = torch.Generator().manual_seed(3)
    w1, b1 = torch.randn(64, 3, 3, 3, generator=g), torch.randn(64, generator=g)
    w2 = torch.randn(64, 64, 1, 1, generator=g)
    p1 = engine_p2.pack_planes_stem2x_weight(w1, b1)
    p2 = engine_p2.pack_planes_stem2x_tail_weight(w2)
    a1 = (p1[0].double() + p1[1].double() / 2048).reshape(2, 2, 2, 32, 8)          # [slab][step][half][co][j]
    patch = torch.randn(3, 10, generator=g).double()       # 3 frame rows x [junk, e0..e8], e = 3 dx + c
    ref0 = torch.einsum('ocrs,rsc->o', w1.double(), patch[:, 1:].reshape(3, 3, 3)) + b1.double()
    frag = torch.zeros(2, 2, 8, dtype=torch.float64)        # [step][half][j]: the gather of csrc/planes_impl.h produce()
    frag[0, 0], frag[0, 1], frag[1, 0] = patch[0, :8], patch[2, :8], patch[1, :8]
    frag[1, 1] = torch.stack([patch[0, 8], patch[0, 9], patch[1, 8], patch[1, 9], patch[2, 8], patch[2, 9], torch.tensor(1.0, dtype=torch.float64),
                              torch.tensor(123.0, dtype=torch.float64)])      # (the pad slot's activation is arbitrary)
    got0 = torch.einsum('tshoj,shj->to', a1, frag).reshape(64)
    assert float((got0 - ref0).abs().max()) <= 2e-5
    # the 1x1: lane (h, pixel) of slab s holds channels 32 s + 8 g + 4 h + e; k-step q = 2 s + u takes g = 2 u, 2 u + 1
    y = torch.randn(64, generator=g).double()
    ref1 = w2.double().reshape(64, 64) @ y
    a2 = (p2[0].double() + p2[1].double() / 2048).reshape(2, 4, 2, 32, 8)          # [slab][q][h'][co][j]
    frag2 = torch.zeros(4, 2, 8, dtype=torch.float64)
    for q in range(4):
        s_, u = divmod(q, 2)
        for hh in range(2):
            for j in range(8):
                gg, e = 2 * u + j // 4, j % 4
                frag2[q, hh, j] = y[32 * s_ + 8 * gg + 4 * hh + e]
    got1 = torch.einsum('tqhoj,qhj->to', a2, frag2).reshape(64)
    assert float((got1 - ref1).abs().max()) <= 2e-5
