"""Parity at the BASELINE.json configurations' OWN shapes (VERDICT r1 "row g"):

  config 2  WIDERFACE_LFD_S   8 x 1920x1080   (P = 43,620 / image)
  config 3  WIDERFACE_LFD_L   1 x 3840x2160   (P = 690,600)
  config 4  TT100K_LFD_L      4 x 1280x720    (one GPU's share; 45 classes + background, softmax, separate towers)

For each: (a) G1 -- fp32 inter-layer storage on the shipped MFMA conv kernels vs the fp32 oracle (<= 1e-4: the math);
(b) G2 -- the fp16 product path vs the fp32 oracle (what decode consumes) and vs the fp16-storage-emulating oracle;
(c) every launch of the product path re-derived in float64 FROM THE TENSORS THE ENGINE STORED (<= 1 fp16 ulp per
launch: no error other than the final rounding enters anywhere), plus the error-growth table per tapped map;
(d) G3 -- decode + threshold + NMS on the ORACLE's logits at those grids: kept candidates index-exact;
(e) fully-convolutional crop consistency of the backbone at 4K.
Measured numbers are written to gpurun_out/parity_fullsize.json (DESIGN.md section 4 quotes them).
"""
import functools
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import net_oracle
from conftest import ROOT
from lfd_amd import configs, engine, engine_g1, ops

pytestmark = pytest.mark.gpu

CASES = {
    'config2': ('WIDERFACE_LFD_S', (8, 1080, 1920), (0, 5)),
    'config3': ('WIDERFACE_LFD_L', (1, 2160, 3840), (0,)),
    'config4': ('TT100K_LFD_L', (4, 720, 1280), (2,)),
}
_REPORT = {}


def _record(key, **kw):
    _REPORT.setdefault(key, {}).update({k: (float(v) if not isinstance(v, (list, dict, str, int)) else v) for k, v in kw.items()})
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    json.dump(_REPORT, open(os.path.join(out, 'parity_fullsize.json'), 'w'), indent=1, sort_keys=True)


@functools.lru_cache(maxsize=None)
def _case(key):
    """model (on the GPU), state_dict copy (CPU), frames NHWC fp16 (CPU), HIP outputs (CPU), oracle outputs per checked image"""
    name, (n, h, w), imgs = CASES[key]
    arch = configs.ARCHS[name]
    m = configs.build_model(name)
    configs.perturb_weights(m)
    m.eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = (torch.rand(n, h, w, 3, generator=torch.Generator().manual_seed(11)) * 2 - 1).half()
    m.cuda()
    with torch.no_grad():
        cls, reg = [t.clone().cpu() for t in m.forward_resident(x.cuda())]
    sizes = [tuple(m.head_indexes_to_feature_map_sizes[i]) for i in range(len(arch['regression_ranges']))]
    ref = {}
    with torch.no_grad():
        for i in imgs:
            xi = x[i:i + 1].float().permute(0, 3, 1, 2).contiguous()
            ref[i] = dict(fp32=net_oracle.lfd_forward(sd, arch, xi), emu=net_oracle.lfd_forward_fp16(sd, arch, xi))
    return dict(name=name, arch=arch, model=m, sd=sd, x=x, cls=cls, reg=reg, sizes=sizes, ref=ref, imgs=imgs)


def _scores(arch, t):
    return t.softmax(-1) if arch['classification_loss_type'] == 'CrossEntropyLoss' else t.sigmoid()


# ------------------------------------------------------------------------------------------------ (a) G1
@pytest.mark.parametrize('name,shape', [('WIDERFACE_LFD_XS', (1, 96, 128)), ('WIDERFACE_LFD_S', (2, 135, 241)),
                                        ('WIDERFACE_LFD_M', (1, 64, 96)), ('WIDERFACE_LFD_L', (1, 100, 156)),
                                        ('TT100K_LFD_S', (1, 64, 64)), ('TT100K_LFD_L', (1, 90, 161)), ('TL_LFD_L', (1, 128, 192)),
                                        ('WIDERFACE_LFD_S', (1, 1080, 1920)), ('TT100K_LFD_L', (1, 720, 1280))])
def test_g1_fp32_storage_on_the_mfma_kernels_matches_the_fp32_oracle(name, shape):
    """Gate G1: <= 1e-4 max-abs on raw cls / reg logits (measured ~1e-5)."""
    arch = configs.ARCHS[name]
    m = configs.build_model(name)
    configs.perturb_weights(m)
    m.eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape[0], 3, shape[1], shape[2], generator=torch.Generator().manual_seed(5)) * 2 - 1
    with torch.no_grad():
        rc, rr, rsizes = net_oracle.lfd_forward(sd, arch, x)
        m.cuda()
        c, r, sizes = engine_g1.lfd_forward_g1(m, x.cuda())
    assert [tuple(s) for s in sizes] == [tuple(s) for s in rsizes]
    ec, er = float((c.cpu() - rc).abs().max()), float((r.cpu() - rr).abs().max())
    print('G1 %s %s: cls %.2e reg %.2e' % (name, shape, ec, er))
    _record('G1 %s %dx%d' % (name, shape[2], shape[1]), cls_max_abs=ec, reg_max_abs=er)
    assert ec <= 1e-4 and er <= 1e-4


# ------------------------------------------------------------------------------------------------ (b) G2
@pytest.mark.parametrize('key', ['config2', 'config3', 'config4'])
def test_g2_fp16_pipeline_vs_fp32_and_emulating_oracles_at_the_baseline_shape(key):
    """fp16 storage / fp32 accumulate at the config's own shape.  Gates (stated, see DESIGN 4: fp16 rounding of weights
    ALONE moves sigma by ~1e-3 on these networks, of activations alone by ~1.5e-3):
      vs fp32 oracle:      sigma(cls) | softmax, sigma(reg) <= 2.5e-3;  raw logits <= 2e-2 (reported)
      vs emulating oracle: raw logits mean <= 1.5e-3, 99.99th percentile <= 7e-3, max <= 1.2e-2  (same rounding points; the
                           remainder is 1-ulp flips seeded by accumulation order, random-walking through ~25 layers.  The
                           MAX over the ~2e5 logits of an image is the extreme of that walk: 0.8e-2 .. 1.05e-2 across kernel
                           revisions that only reorder fp32 sums -- the percentile and the mean do not move)"""
    cs = _case(key)
    arch = cs['arch']
    for i in cs['imgs']:
        c, r = cs['cls'][i], cs['reg'][i]
        rc, rr, rs = cs['ref'][i]['fp32']
        ec, er, es = cs['ref'][i]['emu']
        assert [tuple(s) for s in rs] == cs['sizes']
        raw = max(float((c - rc[0]).abs().max()), float((r - rr[0]).abs().max()))
        sg_c = float((_scores(arch, c) - _scores(arch, rc[0])).abs().max())
        sg_r = float((r.sigmoid() - rr[0].sigmoid()).abs().max())
        emu_max = max(float((c - ec[0]).abs().max()), float((r - er[0]).abs().max()))
        emu_mean = max(float((c - ec[0]).abs().mean()), float((r - er[0]).abs().mean()))
        dall = torch.cat([(c - ec[0]).abs().reshape(-1), (r - er[0]).abs().reshape(-1)]).float().cpu()
        emu_q = float(torch.sort(dall).values[int(0.9999 * (dall.numel() - 1))])
        # the floor: the emulating oracle itself vs the fp32 oracle
        floor_c = float((_scores(arch, ec[0]) - _scores(arch, rc[0])).abs().max())
        floor_r = float((er[0].sigmoid() - rr[0].sigmoid()).abs().max())
        print('%s img %d: vs fp32 raw %.2e score %.2e sigma(reg) %.2e | floor (emulation vs fp32) %.2e %.2e | vs emulation '
              'max %.2e q99.99 %.2e mean %.2e' % (key, i, raw, sg_c, sg_r, floor_c, floor_r, emu_max, emu_q, emu_mean))
        _record('G2 %s img%d' % (key, i), raw_vs_fp32=raw, score_vs_fp32=sg_c, sigma_reg_vs_fp32=sg_r,
                floor_score=floor_c, floor_sigma_reg=floor_r, raw_vs_emulation_max=emu_max, raw_vs_emulation_mean=emu_mean,
                raw_vs_emulation_q9999=emu_q)
        assert raw < 2e-2
        assert sg_c < 2.5e-3 and sg_r < 2.5e-3
        # FROZEN at the round-2 measured envelope (no further widening): the gates proper are the 99.99th percentile
        # and the mean; the max over ~2e5 logits is bounded at the measured extreme (1.05e-2) + 15 %, no longer at 1.5e-2.
        # The mode that meets north_star's 1e-3 is precision='fp32_storage' (tests/test_gpu_precise.py).
        assert emu_q < 7e-3 and emu_mean < 1.5e-3
        assert emu_max < 1.2e-2


# ------------------------------------------------------------------------------------------------ (c) per launch
def _ulp16(v, floor=2.0 ** -10):
    """spacing of fp16 at |v|, floored at the spacing of `floor` (default: an absolute 2^-20 for tiny values, independent
    of how the hardware treats fp16 subnormals)"""
    a = v.abs().clamp(min=floor)
    return torch.exp2(torch.floor(torch.log2(a)) - 10)


def _nchw64(t):
    return t.float().permute(0, 3, 1, 2).double()


@pytest.mark.parametrize('key', ['config2', 'config4'])
def test_every_backbone_launch_rederived_from_the_stored_tensors(key):
    """For one frame of the batch: each conv launch of the plan, recomputed in float64 from the fp16 tensors the engine
    itself stored as that launch's inputs (folded fp16 weights, fp32 bias, residual, ReLU), must equal the stored output up
    to the final fp16 rounding + fp32 accumulation noise (tolerance spelled out in `compare`); >= 99.9 % of the outputs of
    a single-conv launch are bit-identical to the rounded float64 value.  Together with G1 this pins where the end-to-end 1-2e-3 comes from: nowhere but
    the rounding of operands."""
    cs = _case(key)
    m, img = cs['model'], cs['imgs'][0]
    plan = engine.get_plan(m, m._backbone, m._neck, m._head, torch.device('cuda', torch.cuda.current_device()))
    n, h, w = CASES[key][1]
    with torch.no_grad():
        m.forward_resident(cs['x'].cuda())          # leaves every layer's output in the plan's resident buffers
    torch.cuda.synchronize()
    st = plan.state_for(n, h, w)
    rows = []

    def compare(tag, got16, ref64, mag64, flips=0.0, min_exact=0.98):
        """accept |got - ref| <= half an fp16 ulp of the reference (the final rounding)
                               + 2^-19 x sum |x||w|   (fp32 accumulation noise of the MFMA contraction: ~sqrt(K) 2^-24 relative to
                                                       the magnitude of the terms, K <= 1152)
                               + `flips`              (launches that round to fp16 INSIDE -- fused stem 3 x, fused block / tail once --
                                                       carry 1-ulp flips of ~0.5 % of their intermediate values into the output).
        Reported: the worst ratio to that tolerance and the fraction of bit-identical outputs."""
        got = _nchw64(got16)
        tol = 0.5005 * _ulp16(ref64, 2.0 ** -14) + 2.0 ** -19 * mag64 + flips
        d = ((got - ref64).abs() / tol)
        exact = float((got == ref64.float().half().double()).double().mean())
        rows.append((tag, float(d.max()), exact, 1.0, min_exact))
        _record('per-launch %s' % key, rows=[list(r) for r in rows])

    def conv_mag(x, wt, b, **kw):
        return F.conv2d(x.abs(), wt.abs(), b.abs(), **kw)

    # ---- stem (one fused launch for the 'faster' stem, or the first pair for 'fast')
    x64 = cs['x'][img:img + 1].float().permute(0, 3, 1, 2).double()
    y = x64
    first_dst = plan.stem_fused[-1] if plan.stem_fused is not None else plan.stem_out
    nstem = 4 if plan.stem_fused is not None else 2
    flips, mag = 0.0, None
    for li, (k, s, wt, b) in enumerate(plan.stem_ref[:nstem]):
        w64, b64 = wt.cpu().half().double(), b.cpu().double()
        if li:      # an ulp of the incoming intermediate, through this layer's largest weights, a handful of times
            flips = flips * float(w64.abs().sum((1, 2, 3)).max()) + 4 * 2.0 ** -11 * float(y.abs().max()) * float(w64.abs().max())
        mag = conv_mag(y, w64, b64, stride=s, padding=k // 2)
        y = F.conv2d(y, w64, b64, stride=s, padding=k // 2).relu()
        if li + 1 < nstem:
            y = y.float().half().double()
    compare('stem', st.bufs[first_dst][img:img + 1].cpu(), y, mag, flips=flips, min_exact=0.9)
    # ---- every conv launch
    skip = -1
    for ci, c in enumerate(plan.convs):
        if ci == skip:
            continue
        xin = _nchw64(st.bufs[c.src][img:img + 1].cpu())
        w64, b64 = c.ref_w.cpu().half().double(), c.b.cpu().double()
        if c.down is not None and engine._use_fused_down(n, xin.shape[2], xin.shape[3]):
            # the stage's first block ran as ONE launch (csrc/down.hip): conv3x3 s2 -> ReLU -> fp16, the 1x1 s2 branch -> fp16,
            # conv3x3 s1 + branch -> ReLU; y1 and the branch exist only in LDS
            c2 = plan.convs[c.down]
            y1 = F.conv2d(xin, w64, b64, stride=2, padding=1).relu().float().half().double()
            wd, bd = c.ds[3].cpu().half().double(), c.ds[1].cpu().double()
            idn = F.conv2d(xin, wd, bd, stride=2).float().half().double()
            w2, b2 = c2.ref_w.cpu().half().double(), c2.b.cpu().double()
            ref = (F.conv2d(y1, w2, b2, padding=1) + idn).relu()
            mag = conv_mag(y1, w2, b2, padding=1) + idn.abs()
            flips = 4 * 2.0 ** -11 * float(y1.abs().max()) * float(w2.abs().max()) + 2.0 ** -11 * float(idn.abs().max())
            compare('downsample block (3x3 s2 + 1x1 s2 + 3x3) %d->%d @%dx%d' % (c.cin, c.cout, ref.shape[2], ref.shape[3]),
                    st.bufs[c2.dst][img:img + 1].cpu(), ref, mag, flips=flips, min_exact=0.9)
            skip = c.down
            continue
        if c.blk128 is not None and engine._use_fused_block128(n, xin.shape[2], xin.shape[3]):
            # a 128-channel block of the last stage ran as ONE launch (csrc/block128.hip): conv3x3 -> ReLU -> fp16 (LDS only),
            # conv3x3 + input -> ReLU
            c2 = plan.convs[c.blk128]
            y1 = F.conv2d(xin, w64, b64, padding=1).relu().float().half().double()
            w2, b2 = c2.ref_w.cpu().half().double(), c2.b.cpu().double()
            ref = (F.conv2d(y1, w2, b2, padding=1) + xin).relu()
            mag = conv_mag(y1, w2, b2, padding=1) + xin.abs()
            flips = 4 * 2.0 ** -11 * float(y1.abs().max()) * float(w2.abs().max())
            compare('block 2 x conv3x3 s1 128->128 @%dx%d' % (ref.shape[2], ref.shape[3]),
                    st.bufs[c2.dst][img:img + 1].cpu(), ref, mag, flips=flips, min_exact=0.9)
            skip = c.blk128
            continue
        ref = F.conv2d(xin, w64, b64, stride=c.stride, padding=c.ks // 2)
        mag = conv_mag(xin, w64, b64, stride=c.stride, padding=c.ks // 2)
        flips = 0.0
        second = c.tail[3:0:-2] if c.tail is not None else (c.blk[2:0:-1] if c.blk is not None else None)   # (w2 folded, b2)
        if second is not None:       # chained 1x1 tail / fused FasterBlock: conv -> ReLU -> fp16 -> second conv
            mid = ref.relu().float().half().double()
            w2, b2 = second[0].cpu().half().double(), second[1].cpu().double()
            pad2 = w2.shape[-1] // 2
            ref = F.conv2d(mid, w2, b2, padding=pad2)
            mag = conv_mag(mid, w2, b2, padding=pad2)
            flips = 4 * 2.0 ** -11 * float(mid.abs().max()) * float(w2.abs().max())
        if c.res is not None:
            r64 = _nchw64(st.bufs[c.res][img:img + 1].cpu())
            ref = ref + r64
            mag = mag + r64.abs()
        if c.relu:
            ref = ref.relu()
        compare('%s%dx%d s%d %d->%d @%dx%d' % ('block 2 x conv' if c.blk is not None else 'conv', c.ks, c.ks, c.stride, c.cin,
                                              c.cout, ref.shape[2], ref.shape[3]),
                st.bufs[c.dst][img:img + 1].cpu(), ref, mag, flips=flips, min_exact=0.9 if second is not None else 0.98)
        if c.ds is not None:
            wd, bd = c.ds[3].cpu().half().double(), c.ds[1].cpu().double()
            compare('downsample 1x1 s2 %d->%d' % (c.cin, c.cout), st.bufs[c.ds[2]][img:img + 1].cpu(),
                    F.conv2d(xin, wd, bd, stride=2), conv_mag(xin, wd, bd, stride=2))
    # ---- neck + head from the stored taps (emulated rounding points; three fp16 roundings deep)
    taps = [st.bufs[t][img:img + 1].cpu().float().permute(0, 3, 1, 2).contiguous() for t in plan.taps]
    with torch.no_grad():
        hc, hr, _ = net_oracle.head_forward_fp16(cs['sd'], cs['arch'], taps)
    dc = float((cs['cls'][img] - hc[0]).abs().max())
    dr = float((cs['reg'][img] - hr[0]).abs().max())
    bad = [r for r in rows if r[1] > r[3] or r[2] <= r[4]]
    assert not bad, bad
    _record('per-launch %s' % key, worst_ulp=max(r[1] for r in rows[1:]), stem_ulp=rows[0][1],
            min_exact_fraction=min(r[2] for r in rows), head_raw_cls=dc, head_raw_reg=dr, launches=len(rows))
    print('%s: %d launches, worst %.2f x tolerance, stem %.2f, >= %.4f bit-identical; head from stored taps: cls %.2e reg %.2e'
          % (key, len(rows), max(r[1] for r in rows[1:]), rows[0][1], min(r[2] for r in rows), dc, dr))
    assert dc < 4e-3 and dr < 4e-3
    # ---- error growth against the fp32 oracle, per tapped map (relative L2)
    xi = cs['x'][img:img + 1].float().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        ref_taps = net_oracle.backbone_forward(cs['sd'], cs['arch'], xi)
    growth = [float((a - b).norm() / b.norm()) for a, b in zip(taps, ref_taps)]
    _record('per-launch %s' % key, tap_rel_l2_vs_fp32=growth)
    assert max(growth) < 3e-3, growth


# ------------------------------------------------------------------------------------------------ (d) G3
@pytest.mark.parametrize('key', ['config2', 'config3', 'config4'])
@pytest.mark.parametrize('K', [256, 4096])
def test_g3_decode_nms_on_oracle_logits_index_exact_at_the_baseline_grid(key, K):
    """decode + threshold + per-class NMS fed the ORACLE's fp32 logits (fp16-rounded, as SURVEY 8d prescribes) on the
    config's full point grid, threshold = the quantile giving ~K candidates.  (1) the device decode agrees with the
    oracle's decode to fp32 rounding of the transcendental; (2) on the device-decoded boxes/scores the C oracle's
    multiclass_nms and the device pass agree BIT FOR BIT: candidate ordinals (= kept indices, score-descending), labels,
    boxes; (3) the all-oracle pipeline (torch sigmoid instead of expf) keeps the same detections."""
    cs = _case(key)
    arch, m, img = cs['arch'], cs['model'], cs['imgs'][0]
    name, (n, h, w), _ = CASES[key]
    rc, rr, sizes = cs['ref'][img]['fp32']
    cls = rc[0].half().float()
    reg = rr[0].half().float()
    ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
    sc = net_oracle.scores_from_logits(cls.numpy(), ce)
    thr = float(np.partition(sc.reshape(-1), -K)[-K])
    agn = False
    for i, s in enumerate(sizes):
        m._head_indexes_to_feature_map_sizes[i] = tuple(s)
    desc, P = m._detect_desc(thr, 0.4, agn, max_candidates=8192)
    meta = torch.tensor([[float(w), float(h), 1.0]], dtype=torch.float32).cuda()
    out = ops.detect_batched(desc, cls[None].cuda(), reg[None].cuda(), meta)
    boxes, scores = ops.decode_all(desc, cls[None].cuda(), reg[None].cuda(), meta)
    counts = out.counts.cpu().numpy()
    assert counts[0, 2] == 0, 'candidate capacity overflow'
    strides = net_oracle.strides_of(arch)
    # (1)
    rb = net_oracle.decode_boxes(reg.numpy(), [tuple(s) for s in sizes], strides, arch['regression_ranges'], 'sigmoid', 'union', (h, w), 1.0)
    np.testing.assert_allclose(boxes[0].cpu().numpy(), rb, rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(scores[0].cpu().numpy(), sc, rtol=2e-5, atol=1e-7)
    # (2)
    dets, labels, cand, Kc = oracle.multiclass_nms(boxes[0].cpu().numpy(), scores[0].cpu().numpy(), thr, 0.4, agn)
    k = int(counts[0, 1])
    assert counts[0, 0] == Kc and k == len(labels) and k > 0
    np.testing.assert_array_equal(out.cand[0, :k].cpu().numpy(), cand)
    np.testing.assert_array_equal(out.labels[0, :k].cpu().numpy(), labels)
    np.testing.assert_array_equal(out.dets[0, :k].cpu().numpy(), dets)
    # (3) the ALL-ORACLE pipeline (torch.sigmoid scores, torch decode, C NMS) on the same logits, compared by IDENTITY: the kept
    # set as (flat point index, class) pairs.  Only a candidate whose score sits within an ulp of the threshold (expf vs
    # torch.sigmoid) can differ: <= 2 such pairs are tolerated and counted; every common pair must carry the same box.
    odets, olabels, ocand, Ko = net_oracle.get_results_single(cls.numpy(), reg.numpy(), [tuple(s) for s in sizes], strides, arch, thr,
                                                              0.4, agn, (h, w), 1.0)
    osc = net_oracle.scores_from_logits(cls.numpy(), arch['classification_loss_type'] == 'CrossEntropyLoss')
    oflat = np.flatnonzero(osc.reshape(-1) > np.float32(thr))
    assert len(oflat) == Ko
    ncls = osc.shape[1]
    okeys = {(int(oflat[c] // ncls), int(l)): row for c, l, row in zip(ocand, olabels, odets)}
    a = out.dets[0, :k].cpu().numpy()
    la = out.labels[0, :k].cpu().numpy()
    pa = out.point[0, :k].cpu().numpy()
    dkeys = {(int(p_), int(l)): row for p_, l, row in zip(pa, la, a)}
    assert len(dkeys) == k and len(okeys) == len(olabels)
    only = set(dkeys) ^ set(okeys)
    # gated at the count measured on the MI355X in rounds 3-5 for all six (config, K) cases: none (rounds 3-5 tolerated 2)
    assert len(only) == 0, sorted(only)[:8]
    matched = 0
    for key_ in set(dkeys) & set(okeys):
        r1, r2 = dkeys[key_], okeys[key_]
        assert np.abs(r1[:4] - r2[:4]).max() < 5e-4 and abs(r1[4] - r2[4]) < 1e-6 + 2e-5 * r2[4], (key_, r1, r2)
        matched += 1
    _record('G3 %s K=%d' % (key, K), candidates=int(Kc), kept=k, all_oracle_kept=len(olabels), matched=matched, points=int(P),
            kept_only_on_one_side=len(only))


@pytest.mark.parametrize('key,K', [('config2', 256), ('config2', 4096), ('config4', 256)])
def test_fp32_storage_end_to_end_kept_list_is_the_oracle_pipelines_list_in_order(key, K):
    """north_star: "bit-exact box indices after NMS".  The WHOLE product path in the tolerance-compliant mode -- its own
    'fp32_storage' forward on the frame + device decode / threshold / NMS (lfd.py:434-509, nms.py:161-220) -- against the
    all-oracle pipeline on the fp32 oracle's logits of the same frame (oracle/net_oracle.py, pinned to the reference to 2e-5):
    the kept detections as an ORDERED list of (flat point index, class) must be the same list.  The two forwards differ by
    <= 8e-6 in the raw logits, so a score within that distance of the threshold, or two kept scores closer than that, could
    legitimately differ: counted, recorded, and gated at the counts measured on the MI355X (zero)."""
    cs = _case(key)
    arch, m, img = cs['arch'], cs['model'], cs['imgs'][0]
    name, (n, h, w), _ = CASES[key]
    rc, rr, sizes = cs['ref'][img]['fp32']
    ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
    osc = net_oracle.scores_from_logits(rc[0].numpy(), ce)
    thr = float(np.partition(osc.reshape(-1), -K)[-K])
    strides = net_oracle.strides_of(arch)
    iou = 0.4 if not ce else 0.1
    odets, olabels, ocand, Ko = net_oracle.get_results_single(rc[0].numpy(), rr[0].numpy(), [tuple(s) for s in sizes], strides, arch, thr,
                                                              iou, False, (h, w), 1.0)
    oflat = np.flatnonzero(osc.reshape(-1) > np.float32(thr))
    ncls = osc.shape[1]
    olist = [(int(oflat[c] // ncls), int(l)) for c, l in zip(ocand, olabels)]
    keep = (m.precision, m._classification_threshold, dict(m._nms_cfg) if m._nms_cfg else None, m.use_graph)
    try:
        m.precision = 'fp32_storage'
        m._classification_threshold, m._nms_cfg, m.use_graph = thr, dict(type='nms', iou_thr=iou), False
        m.max_candidates = 8192
        meta = torch.tensor([[float(w), float(h), 1.0]], dtype=torch.float32).cuda()
        with torch.no_grad():
            out = m.detect_resident(cs['x'][img:img + 1].cuda(), meta)
        counts = out.counts.cpu().numpy()
        assert counts[0, 2] == 0, 'candidate capacity overflow'
        k = int(counts[0, 1])
        dlist = [(int(p_), int(l)) for p_, l in zip(out.point[0, :k].cpu().numpy(), out.labels[0, :k].cpu().numpy())]
        dd = out.dets[0, :k].cpu().numpy()
    finally:
        m.precision, m._classification_threshold, m._nms_cfg, m.use_graph = keep
    only = set(dlist) ^ set(olist)
    order = sum(1 for a_, b_ in zip(dlist, olist) if a_ != b_) if not only else None
    print('%s K=%d: oracle keeps %d, the product %d; on one side only %d; positions that differ %s' % (key, K, len(olist), len(dlist), len(only), order))
    _record('identity %s K=%d' % (key, K), oracle_kept=len(olist), product_kept=len(dlist), only_on_one_side=len(only), positions_that_differ=order)
    assert len(olist) > 0 and len(only) == 0, sorted(only)[:8]
    # the SET is identical.  multiclass_nms orders the kept rows by score (nms.py:215-218): a position may only differ where the
    # oracle's own scores of the two rows involved are closer than the two forwards are to each other (measured on the MI355X:
    # config 2: K = 256 one adjacent swap = 2 positions of 243, K = 4096 15 swaps = 30 positions of 3615) -- anything else is an error;
    # the count is bounded at 1 % of the kept rows
    oscore = {key_: float(r[4]) for key_, r in zip(olist, odets)}
    for i, (a_, b_) in enumerate(zip(dlist, olist)):
        if a_ != b_:
            assert abs(oscore[a_] - oscore[b_]) < 2e-5, (i, a_, b_, oscore[a_], oscore[b_])
    assert order <= max(4, len(olist) // 100), order
    orow = {key_: r for key_, r in zip(olist, odets)}
    for key_, r1 in zip(dlist, dd):
        r2 = orow[key_]
        assert np.abs(r1[:4] - r2[:4]).max() < 2e-3 and abs(r1[4] - r2[4]) < 2e-5, (key_, r1, r2)


# ------------------------------------------------------------------------------------------------ (e) crop consistency
def test_config3_backbone_is_fully_convolutional_at_4k():
    """WIDERFACE_LFD_L at 3840x2160: the top-left 1920x1080 crop run ALONE reproduces the tapped maps of the full frame
    bit for bit wherever the receptive field does not reach the crop's right / bottom edge (BatchNorm is folded, so the
    backbone is purely convolutional; tiles are anchored at the top-left corner, so the MFMA accumulation order of a pixel
    does not depend on the frame size).  The head is NOT crop-consistent by construction (GroupNorm statistics run over
    the whole level), which is why this is asserted on the backbone."""
    cs = _case('config3')
    m, arch = cs['model'], cs['arch']
    bb = m._backbone
    x = cs['x'].cuda()
    with torch.no_grad():
        full = [t.clone() for t in bb(x)]
        crop = [t.clone() for t in bb(x[:, :1080, :1920].contiguous())]
    # receptive-field radius (input pixels) of every tapped map: r += (k-1)/2 * jump, jump *= stride
    radius, jump, radii = 0, 1, []
    seq = [(3, 2), (1, 1)] if arch['stem_mode'] == 'fast' else [(3, 2), (1, 1), (3, 2), (1, 1)]
    for k, s in seq:
        radius += (k // 2) * jump
        jump *= s
    taps = sorted(tuple(t) for t in arch['out_indices'])
    for i, nblk in enumerate(arch['body_architecture']):
        for j in range(nblk):
            s = 2 if j == 0 else 1
            radius += 1 * jump          # conv1 3x3 (stride s)
            jump *= s
            radius += 1 * jump          # conv2 3x3 s1
            if (i, j) in taps:
                radii.append((radius, jump))
    assert len(radii) == len(full)
    checked = 0
    for (r, jp), f, c in zip(radii, full, crop):
        hh = (1080 - r) // jp
        ww = (1920 - r) // jp
        if hh <= 0 or ww <= 0:
            continue
        assert torch.equal(f[:, :, :hh, :ww], c[:, :, :hh, :ww]), (r, jp)
        checked += 1
        # and the border region really differs (the check is not vacuous)
        assert not torch.equal(f[:, :, :c.shape[2], :c.shape[3]], c)
    assert checked >= 3


# ------------------------------------------------------------------------------------------------ config 4 end to end
def test_config4_multiclass_end_to_end_is_index_exact():
    """TT100K_LFD_L, 4 x 1280x720, 45 classes: LFD.get_results on the engine's own logits == the C oracle's
    multiclass_nms on the device-decoded boxes / softmax scores, for every image, detection for detection (labels, kept
    order, coordinates bit for bit incl. the class-offset (b+off)-off rounding) -- a count mismatch FAILS."""
    cs = _case('config4')
    m = cs['model']
    cls, reg = cs['cls'].cuda(), cs['reg'].cuda()
    for i, s in enumerate(cs['sizes']):
        m._head_indexes_to_feature_map_sizes[i] = tuple(s)
    sc = cls[0].softmax(-1)[:, :-1]
    for q, iou, agn in ((2e-4, 0.1, False), (2e-3, 0.1, True), (1e-4, 0.4, False)):
        thr = float(torch.quantile(sc.reshape(-1)[::7].float(), 1 - q))
        m._classification_threshold = thr
        m._nms_cfg = dict(type='nms', iou_thr=iou)
        if agn:
            m._nms_cfg['class_agnostic'] = True
        meta_l = [dict(resized_height=720, resized_width=1280, resize_scale=1.0)] * 4
        res = m.get_results((cls, reg), meta_l)
        desc, _ = m._detect_desc(thr, iou, agn)
        meta = torch.tensor([[1280.0, 720.0, 1.0]] * 4).cuda()
        boxes, scores = ops.decode_all(desc, cls, reg, meta)
        for n in range(4):
            dets, labels, _, K = oracle.multiclass_nms(boxes[n].cpu().numpy(), scores[n].cpu().numpy(), thr, iou, agn)
            ref = net_oracle.pack_results(dets, labels)
            assert K > 50
            assert len(ref) == len(res[n]), (n, len(ref), len(res[n]))
            assert [r[0] for r in res[n]] == [r[0] for r in ref]
            np.testing.assert_array_equal(np.array(res[n], np.float32), np.array(ref, np.float32))


# ------------------------------------------------------------------------------------------------ stale graphs
def test_whole_step_graph_follows_parameter_updates():
    """detect_resident(use_graph=True) -> load_state_dict -> detect_resident: the replayed step must use the NEW weights
    (the graph cache is dropped when the engine plan is rebuilt)."""
    m = configs.build_model('WIDERFACE_LFD_XS')
    configs.perturb_weights(m)
    m.eval().cuda()
    m.use_graph = True
    x = (torch.rand(2, 96, 128, 3, device='cuda') * 2 - 1).half()
    meta = torch.tensor([[128.0, 96.0, 1.0]] * 2).cuda()
    with torch.no_grad():
        a = m.detect_resident(x, meta, score_thr=0.3)
        a_dets, a_counts = a.dets.clone(), a.counts.clone()
        cls_a = m.forward_resident(x)[0].clone()
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        other = configs.build_model('WIDERFACE_LFD_XS', seed=7)
        configs.perturb_weights(other, seed=9)
        m.load_state_dict(other.state_dict())
        b = m.detect_resident(x, meta, score_thr=0.3)
        b_dets, b_counts = b.dets.clone(), b.counts.clone()
        cls_b = m.forward_resident(x)[0].clone()
        assert not torch.equal(cls_a, cls_b)
        assert not (torch.equal(a_dets, b_dets) and torch.equal(a_counts, b_counts))
        m.load_state_dict(sd)                      # back to the first weights: same results as the first call
        c = m.detect_resident(x, meta, score_thr=0.3)
        assert torch.equal(c.counts, a_counts)
        k = int(a_counts[0, 1])
        assert torch.equal(c.dets[0, :k], a_dets[0, :k])
