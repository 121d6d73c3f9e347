"""The SHIPPED 'fp32_storage' precision mode (LFD.precision = 'fp32_storage'; lfd_amd/engine_p32.py, csrc/precise.hip)
against north_star's tolerance -- "cls/bbox tensors within 1e-3" of the reference's fp32 path (lfd/model/lfd.py:511-542):

  * kernels: lfd_p32_conv2d_nhwc_f32 (every ks / stride / channel-count class, residual, ReLU, Scale, strided outputs, the
    three frame formats of the first conv) and lfd_p32_groupnorm_relu_f32 vs float64 PyTorch of the same op;
  * whole networks: every named configuration at small shapes and BASELINE configs 2 / 3 / 4 at their OWN shapes vs the fp32
    oracle (oracle/net_oracle.py, pinned to the reference by tests/golden): gates raw logits <= 1e-4 AND sigma / softmax
    <= 1e-3 (measured ~1e-5); results of get_results identical to the oracle pipeline on the oracle's logits;
  * LFD API: forward / get_results / detect_resident (+ HIP graph) in that mode; bit-identical replay; mode switch.
Measured numbers go to gpurun_out/parity_precise.json (DESIGN 4 quotes them).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net_oracle
from conftest import ROOT
from lfd_amd import _lib, configs, engine_p32
from lfd_amd._lib import check, lib, ptr, stream_ptr

pytestmark = pytest.mark.gpu
_REPORT = {}


def _record(key, **kw):
    _REPORT.setdefault(key, {}).update({k: float(v) for k, v in kw.items()})
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    json.dump(_REPORT, open(os.path.join(out, 'parity_precise.json'), 'w'), indent=1, sort_keys=True)


def _conv(x, w, b, ks, stride, relu, res=None, scale=None, out=None, pix_stride=0, img_stride=0, fmt=-1):
    """x: NHWC fp32 cuda (fmt < 0) or a frame batch in `fmt`; w [cout, cin, ks, ks] fp32"""
    dev = torch.device('cuda')
    if fmt >= 0:
        n, h, wd = (x.shape[0], x.shape[2], x.shape[3]) if fmt == 0 else (x.shape[0], x.shape[1], x.shape[2])
        cin = 3
        w27 = w.permute(0, 2, 3, 1).reshape(w.shape[0], 27)
        wp = engine_p32.pack_weight(torch.cat([w27, w27.new_zeros((w.shape[0], 5))], 1).reshape(w.shape[0], 32, 1, 1))
    else:
        n, h, wd, cin = x.shape
        wp = engine_p32.pack_weight(w)
    cout = w.shape[0]
    oh, ow = (h + 2 * (ks // 2) - ks) // stride + 1, (wd + 2 * (ks // 2) - ks) // stride + 1
    if out is None:
        out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=dev)
    d = _lib.P32ConvDesc(n, h, wd, cin, cout, ks, stride, int(relu), fmt, pix_stride, img_stride)
    wd_, bd_ = wp.to(dev), engine_p32._pad_bias(b).to(dev)     # named: a temporary would be freed (and its block reused) before the launch runs
    check(lib().lfd_p32_conv2d_nhwc_f32(C.byref(d), ptr(x), ptr(out), ptr(wd_), ptr(bd_), ptr(res), ptr(scale), stream_ptr()),
          'lfd_p32_conv2d_nhwc_f32')
    torch.cuda.synchronize()
    return out


def _ref_conv(x_nhwc, w, b, ks, stride, relu, res=None, scale=None):
    y = F.conv2d(x_nhwc.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double()
    if scale is not None:
        y = y * float(scale)
    return y.relu() if relu else y


@pytest.mark.parametrize('cin,cout,ks,stride,h,w', [
    (64, 64, 3, 1, 19, 37), (64, 64, 3, 2, 33, 50), (64, 64, 1, 1, 9, 21), (64, 64, 1, 2, 17, 31), (64, 128, 3, 2, 34, 60),
    (128, 128, 3, 1, 17, 30), (128, 128, 1, 1, 17, 30), (32, 64, 3, 2, 20, 28), (32, 32, 1, 1, 8, 16), (64, 128, 1, 1, 16, 16),
    (128, 5, 1, 1, 11, 13), (128, 46, 1, 1, 7, 9), (128, 4, 3, 1, 6, 10)])
def test_p32_conv_vs_float64(cin, cout, ks, stride, h, w):
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + ks + stride)
    x = torch.randn(2, h, w, cin, generator=g) * 2
    wt = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)
    b = torch.randn(cout, generator=g)
    ref = _ref_conv(x, wt, b, ks, stride, True)
    got = _conv(x.cuda(), wt, b, ks, stride, True)
    assert got.shape == ref.shape
    err = float((got.cpu().double() - ref).abs().max())
    mag = float(ref.abs().max())
    print('p32 conv %d->%d k%d s%d: err %.2e (max |y| %.2f)' % (cin, cout, ks, stride, err, mag))
    assert err <= 3e-6 * max(1.0, mag)
    # residual + no ReLU + Scale, on the same shapes
    res = torch.randn(ref.shape, generator=g)
    sc = torch.tensor([1.37])
    ref2 = _ref_conv(x, wt, b, ks, stride, False, res, sc)
    xd, resd, scd = x.cuda(), res.cuda(), sc.cuda()
    got2 = _conv(xd, wt, b, ks, stride, False, res=resd, scale=scd)
    assert float((got2.cpu().double() - ref2).abs().max()) <= 3e-6 * max(1.0, float(ref2.abs().max()))


def test_p32_conv_writes_into_a_level_concatenated_output():
    """the head's output convs write straight into [N,P,C'] at a point offset (lfd.py:526-542 builds that layout with
    permute + reshape + cat); everything outside the level's rows and channels stays untouched"""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 7, 128, generator=g)
    wt, b = torch.randn(4, 128, 1, 1, generator=g) * 0.1, torch.randn(4, generator=g)
    P, off = 100, 20
    out = torch.full((3, P, 4), -7.0, device='cuda')
    view = out.view(-1)[off * 4:]
    _conv(x.cuda(), wt, b, 1, 1, False, out=view, pix_stride=4, img_stride=P * 4)
    ref = _ref_conv(x, wt, b, 1, 1, False).reshape(3, 35, 4)
    o = out.cpu()
    assert float((o[:, off:off + 35].double() - ref).abs().max()) < 1e-5
    assert bool((o[:, :off] == -7).all()) and bool((o[:, off + 35:] == -7).all())
    # odd channel count (46 classes + a stride that is not a multiple of 4): scalar store path
    wt, b = torch.randn(45, 128, 1, 1, generator=g) * 0.1, torch.randn(45, generator=g)
    out = torch.full((3, P, 45), -7.0, device='cuda')
    _conv(x.cuda(), wt, b, 1, 1, False, out=out.view(-1)[off * 45:], pix_stride=45, img_stride=P * 45)
    ref = _ref_conv(x, wt, b, 1, 1, False).reshape(3, 35, 45)
    o = out.cpu()
    assert float((o[:, off:off + 35].double() - ref).abs().max()) < 1e-5
    assert bool((o[:, :off] == -7).all()) and bool((o[:, off + 35:] == -7).all())


@pytest.mark.parametrize('fmt', [0, 1, 2])
def test_p32_first_conv_gathers_its_patches_from_the_frame(fmt):
    """3x3 stride-2 conv on the 3-channel frame, formats NCHW fp32 / NHWC fp16 / NHWC uint8 + simple_normalize
    (augmentation_pipeline.py:31-36: (x / 255 - 0.5) / 0.5)"""
    g = torch.Generator().manual_seed(fmt)
    n, h, w = 2, 37, 53
    wt, b = torch.randn(64, 3, 3, 3, generator=g) * 0.3, torch.randn(64, generator=g)
    if fmt == 0:
        x = torch.rand(n, 3, h, w, generator=g) * 2 - 1
        xr = x.permute(0, 2, 3, 1)
    elif fmt == 1:
        x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half()
        xr = x.float()
    else:
        x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
        xr = (x.float() / 255 - 0.5) / 0.5
    ref = _ref_conv(xr, wt, b, 3, 2, True)
    got = _conv(x.cuda(), wt, b, 3, 2, True, fmt=fmt)
    assert got.shape == ref.shape
    assert float((got.cpu().double() - ref).abs().max()) <= 3e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('n,hw,relu', [(2, 17 * 30, 1), (1, 135 * 240, 1), (3, 5, 0)])
def test_p32_groupnorm_relu_vs_float64(n, hw, relu):
    g = torch.Generator().manual_seed(hw)
    x = torch.randn(n, hw, 128, generator=g) * 3 + 0.7
    gamma, beta = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    ref = F.group_norm(x.double().permute(0, 2, 1), 16, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1)
    if relu:
        ref = ref.relu()
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
    ws = torch.empty(int(lib().lfd_p32_groupnorm_workspace_bytes(n, 16)), dtype=torch.uint8, device='cuda')
    check(lib().lfd_p32_groupnorm_relu_f32(ptr(xd), n, hw, 128, 16, ptr(gd), ptr(bd), 1e-5, relu, ptr(ws), ws.numel(), stream_ptr()),
          'lfd_p32_groupnorm_relu_f32')
    assert float((xd.cpu().double() - ref).abs().max()) <= 5e-6


def _model(name):
    m = configs.build_model(name)
    configs.perturb_weights(m)
    m.eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m, sd


def _scores(arch, t):
    return t.softmax(-1) if arch['classification_loss_type'] == 'CrossEntropyLoss' else t.sigmoid()


def _gate(tag, arch, c, r, rc, rr):
    raw = max(float((c - rc).abs().max()), float((r - rr).abs().max()))
    sg = max(float((_scores(arch, c) - _scores(arch, rc)).abs().max()), float((r.sigmoid() - rr.sigmoid()).abs().max()))
    print('%s: raw %.2e sigma %.2e' % (tag, raw, sg))
    _record(tag, raw_max_abs=raw, sigma_max_abs=sg)
    assert raw <= 1e-4, 'raw logits: %g' % raw          # VERDICT r2 task 1: raw <= 1e-4 ...
    assert sg <= 1e-3, 'sigma / softmax: %g' % sg        # ... and north_star's "within 1e-3" on what decode consumes


@pytest.mark.parametrize('name,shape', [('WIDERFACE_LFD_XS', (1, 96, 128)), ('WIDERFACE_LFD_S', (2, 135, 241)),
                                        ('WIDERFACE_LFD_M', (1, 64, 96)), ('WIDERFACE_LFD_L', (1, 100, 156)),
                                        ('TT100K_LFD_S', (1, 64, 64)), ('TT100K_LFD_L', (2, 90, 161)), ('TL_LFD_L', (1, 128, 192)),
                                        ('TL_LFD_S', (1, 96, 160))])
def test_precise_mode_every_configuration_vs_fp32_oracle(name, shape):
    arch = configs.ARCHS[name]
    m, sd = _model(name)
    x = torch.rand(shape[0], 3, shape[1], shape[2], generator=torch.Generator().manual_seed(5)) * 2 - 1
    with torch.no_grad():
        rc, rr, rsizes = net_oracle.lfd_forward(sd, arch, x)
        m.cuda()
        m.precision = 'fp32_storage'
        c, r = m(x.cuda())
    assert [tuple(m.head_indexes_to_feature_map_sizes[i]) for i in range(len(rsizes))] == [tuple(s) for s in rsizes]
    _gate('small %s %dx%d' % (name, shape[2], shape[1]), arch, c.cpu(), r.cpu(), rc, rr)


@pytest.mark.parametrize('key,name,shape,imgs', [('config2', 'WIDERFACE_LFD_S', (8, 1080, 1920), (0, 5)),
                                                 ('config3', 'WIDERFACE_LFD_L', (1, 2160, 3840), (0,)),
                                                 ('config4', 'TT100K_LFD_L', (4, 720, 1280), (2,))])
def test_precise_mode_at_the_baseline_configs_own_shapes(key, name, shape, imgs):
    """BASELINE.json configs 2 / 3 / 4 at their own shapes, NHWC fp16 frames resident on the device (the bench's input
    format): raw <= 1e-4 and sigma / softmax <= 1e-3 vs the fp32 oracle for the checked images of the batch; the mode's
    detections == the oracle pipeline's on the oracle's logits (same boxes to 1e-3 px, same count)."""
    arch = configs.ARCHS[name]
    m, sd = _model(name)
    n, h, w = shape
    x = (torch.rand(n, h, w, 3, generator=torch.Generator().manual_seed(11)) * 2 - 1).half()
    m.cuda()
    m.precision = 'fp32_storage'
    with torch.no_grad():
        c, r = m.forward_resident(x.cuda())
        c, r = c.cpu(), r.cpu()
    sizes = [tuple(m.head_indexes_to_feature_map_sizes[i]) for i in range(len(arch['regression_ranges']))]
    for i in imgs:
        xi = x[i:i + 1].float().permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            rc, rr, rs = net_oracle.lfd_forward(sd, arch, xi)
        assert [tuple(s) for s in rs] == sizes
        _gate('%s img%d' % (key, i), arch, c[i], r[i], rc[0], rr[0])
        # detections through the product's get_results in this mode vs the oracle pipeline on the ORACLE's logits
        sc = _scores(arch, rc[0])
        sc = sc[:, :-1] if arch['classification_loss_type'] == 'CrossEntropyLoss' else sc
        thr = float(np.quantile(sc.max(-1).values.numpy(), 1 - 300.0 / sc.shape[0]))
        m._classification_threshold = thr
        meta = [dict(resized_height=h, resized_width=w, resize_scale=1.0)]
        got = m.get_results((c[i:i + 1].cuda(), r[i:i + 1].cuda()), meta)[0]
        dets, labels, _, _ = net_oracle.get_results_single(rc[0].numpy(), rr[0].numpy(), sizes, net_oracle.strides_of(arch), arch,
                                                           thr, 0.4, False, (h, w), 1.0)
        ref = net_oracle.pack_results(dets, labels)
        # a candidate whose score is within 1e-5 of the threshold, or an IoU within 1e-5 of 0.4, may legitimately differ; rows are
        # matched by content (two scores closer than 1e-5 may swap places in the score-descending order)
        assert abs(len(got) - len(ref)) <= 2 and len(ref) > 50, (len(got), len(ref))
        a, b = np.array(got, np.float64), np.array(ref, np.float64)
        matched = 0
        for row in b:
            d = np.abs(a[:, 2:] - row[2:]).max(1) + 1e3 * (a[:, 0] != row[0]) + 1e3 * (np.abs(a[:, 1] - row[1]) > 1e-4)
            matched += int(d.min() < 1e-2)
        assert matched >= len(b) - 2, (matched, len(b))


def test_precise_mode_uint8_frames_equal_their_normalised_fp32_frames():
    """resident uint8 NHWC frames (simple_normalize inside the fused stem, augmentation_pipeline.py:31-36 -- the row-stream stem
    kernel looks the bytes up in a table of split normalised values) against the SAME frames normalised on the host and fed as
    fp32 NCHW (the two-launch stem with its loaders): the logits agree like two orders of fp32 sums do, and sit inside the mode's gate
    against the fp32 oracle"""
    name = 'WIDERFACE_LFD_S'
    arch = configs.ARCHS[name]
    m, sd = _model(name)
    m.cuda()
    m.precision = 'fp32_storage'
    g = torch.Generator().manual_seed(11)
    x8 = torch.randint(0, 256, (2, 136, 240, 3), generator=g, dtype=torch.uint8)
    xf = ((x8.float() / 255 - 0.5) / 0.5).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        rc, rr, _ = net_oracle.lfd_forward(sd, arch, xf)
        c8, r8 = [t.clone() for t in m.forward_resident(x8.cuda())]
        cf, rf = m(xf.cuda())
    assert float((c8 - cf).abs().max()) <= 2e-5 and float((r8 - rf).abs().max()) <= 2e-5
    _gate('uint8 frames %s' % name, arch, c8.cpu(), r8.cpu(), rc, rr)


def test_precise_mode_api_graph_replay_and_mode_switch():
    """detect_resident in precise mode (one HIP graph per step) == the eager precise step, bit for bit, twice; switching the
    mode back gives the fp16 engine's outputs again; both differ from each other by the fp16 rounding (sanity: the switch
    really switches)."""
    name = 'WIDERFACE_LFD_S'
    m, _ = _model(name)
    m.cuda()
    x = (torch.rand(2, 270, 480, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1).half().cuda()
    meta = torch.tensor([[480., 270., 1.0]] * 2, device='cuda')
    with torch.no_grad():
        c16, r16 = [t.clone() for t in m.forward_resident(x)]
        m.precision = 'fp32_storage'
        c32, r32 = [t.clone() for t in m.forward_resident(x)]
        thr = float(c32.sigmoid().flatten().kthvalue(c32.numel() - 200).values)
        eager = m.detect_resident(x, meta, score_thr=thr, iou_thr=0.4)
        ed, ec = eager.dets.clone(), eager.counts.clone()
        m.use_graph = True
        for _ in range(2):
            out = m.detect_resident(x, meta, score_thr=thr, iou_thr=0.4)
            torch.cuda.synchronize()
            assert torch.equal(out.counts, ec)
            for i in range(2):
                k = int(ec[i, 1])
                assert k > 0 and torch.equal(out.dets[i, :k], ed[i, :k])
        cg, rg = m.forward_resident(x)
        assert torch.equal(cg, c32) and torch.equal(rg, r32)
        m.use_graph = False
        m.precision = 'fp16'
        c16b, r16b = m.forward_resident(x)
        assert torch.equal(c16b, c16) and torch.equal(r16b, r16)
    d = float((c16 - c32).abs().max())
    assert 1e-5 < d < 5e-2, d


@pytest.mark.parametrize('fmt', [-1, 1])
def test_p32_conv_with_chained_1x1_keeps_the_intermediate_in_lds(fmt):
    """lfd_p32_conv2d_tail_nhwc_f32: a stem pair conv3x3 s2 + ReLU -> conv1x1 (64 -> 64) + ReLU in one launch (the first pair
    reads the frame, the second fp32 NHWC maps) vs the float64 chain; odd sizes (tile overhang in both directions)."""
    g = torch.Generator().manual_seed(7 + fmt)
    n, h, w = 2, 37, 51
    if fmt < 0:
        cin = 64
        x = torch.randn(n, h, w, cin, generator=g)
        xr = x
        wt = torch.randn(64, cin, 3, 3, generator=g) / 24
        wp = engine_p32.pack_weight(wt)
    else:
        cin = 3
        x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half()
        xr = x.float()
        wt = torch.randn(64, 3, 3, 3, generator=g) * 0.3
        w27 = wt.permute(0, 2, 3, 1).reshape(64, 27)
        wp = engine_p32.pack_weight(torch.cat([w27, w27.new_zeros((64, 5))], 1).reshape(64, 32, 1, 1))
    b = torch.randn(64, generator=g)
    w2, b2 = torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g)
    ref = _ref_conv(_ref_conv(xr, wt, b, 3, 2, True).float(), w2, b2, 1, 1, True)       # fp32 hand-off like the kernel's LDS planes
    ref64 = _ref_conv(_ref_conv(xr, wt, b, 3, 2, True), w2, b2, 1, 1, True)
    oh, ow = ref.shape[1], ref.shape[2]
    out = torch.empty((n, oh, ow, 64), dtype=torch.float32, device='cuda')
    xd, wd, bd = x.cuda(), wp.cuda(), engine_p32._pad_bias(b).cuda()
    w2d, b2d = engine_p32.pack_weight(w2).cuda(), engine_p32._pad_bias(b2).cuda()
    d = _lib.P32ConvDesc(n, h, w, cin, 64, 3, 2, 1, fmt, 0, 0)
    check(lib().lfd_p32_conv2d_tail_nhwc_f32(C.byref(d), ptr(xd), ptr(out), ptr(wd), ptr(bd), ptr(w2d), ptr(b2d), 1, stream_ptr()),
          'lfd_p32_conv2d_tail_nhwc_f32')
    torch.cuda.synchronize()
    err = float((out.cpu().double() - ref64).abs().max())
    assert err <= 4e-6 * max(1.0, float(ref64.abs().max())), err
    # the chained launch == the two separate launches to fp32 rounding of the hand-off
    mid = _conv(xd, wt, b, 3, 2, True, fmt=fmt)
    two = _conv(mid, w2, b2, 1, 1, True)
    assert float((two - out).abs().max()) <= 2e-6 * max(1.0, float(ref64.abs().max()))
