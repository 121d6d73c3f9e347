"""End to end (G4): engine forward + fused post-processing vs the reference's get_results on its
own fp32 forward.  The forwards differ by fp16 storage, so detections are matched by IoU/score;
where a score sits within the forward tolerance of the threshold the candidate set may differ --
those are counted and bounded, not hidden."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from lfd_amd import configs

pytestmark = pytest.mark.gpu


# Unmatched detections (reference rows without a partner + rows of ours without one), gated at the count MEASURED on the MI355X
# + 1 (round 4, gpurun_out/end2end_counts.json; rounds 1-3 allowed 10 % of the reference's rows).  A candidate whose score sits
# within the fp16 forward's tolerance of the threshold can fall on the other side of it: that is what these counts are.
# Matching is SURVEY 8d G4's: same class, IoU >= 0.99, |score difference| <= 2e-3.  Two modes: 'fp16' (the headline mode:
# fp16 storage against the reference's fp32) and 'fp32_storage' (the mode inside north_star's tolerance: the reference's rows
# must come back essentially row for row).
_IOU, _DSCORE = 0.99, 2e-3
# Unmatched detections (reference rows without a partner + rows of ours without one): gates = the counts MEASURED on the MI355X
# (round 5, gpurun_out/end2end_counts.json) + 1 in 'fp16' mode -- a candidate whose score sits within the fp16 forward's
# tolerance of the threshold can fall on the other side of it -- and the measured counts themselves, ZERO, in 'fp32_storage'.
_UNMATCHED_GATE = {('end_to_end/WIDERFACE_LFD_XS', 'fp16'): 2,          # measured 1 of 43
                   ('end_to_end/WIDERFACE_LFD_S', 'fp16'): 1,           # 0 of 21
                   ('config1/predict_py', 'fp16'): 6,                   # 5 of 1148 (thr 0.5 / IoU 0.3, predict.py:22)
                   ('config1/q90', 'fp16'): 1,                          # 0 of 314
                   ('end_to_end/WIDERFACE_LFD_XS', 'fp32_storage'): 0,  # 0 of 43
                   ('end_to_end/WIDERFACE_LFD_S', 'fp32_storage'): 0,   # 0 of 21
                   ('config1/predict_py', 'fp32_storage'): 0,           # 0 of 1148 (VERDICT r4 asked <= 1)
                   ('config1/q90', 'fp32_storage'): 0}                  # 0 of 314


# rows of the 'fp32_storage' result that are not the reference's row at the same position: the measured counts
_ORDER_GATE = {'config1/predict_py': 0, 'config1/q90': 0}


def _record(key, value):
    from conftest import ROOT
    d = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(d, exist_ok=True)
    f = os.path.join(d, 'end2end_counts.json')
    try:
        cur = json.load(open(f))
    except Exception:
        cur = {}
    cur[key] = value
    json.dump(cur, open(f, 'w'), indent=1)


def _iou(a, b):
    ax2, ay2, bx2, by2 = a[0] + a[2] - 1, a[1] + a[3] - 1, b[0] + b[2] - 1, b[1] + b[3] - 1
    w = max(0.0, min(ax2, bx2) - max(a[0], b[0]))
    h = max(0.0, min(ay2, by2) - max(a[1], b[1]))
    u = (ax2 - a[0]) * (ay2 - a[1]) + (bx2 - b[0]) * (by2 - b[1]) - w * h
    return w * h / u if u > 0 else 1.0


@pytest.mark.parametrize('precision', ['fp16', 'fp32_storage'])
@pytest.mark.parametrize('name', ['WIDERFACE_LFD_XS', 'WIDERFACE_LFD_S'])
def test_end_to_end_detections_match_reference(name, precision):
    g = load_golden('ref_model_%s.npz' % name)
    m = configs.build_model(name, seed=666)
    configs.perturb_weights(m, seed=1)
    m.eval().cuda()
    m.precision = precision
    N, H, W = [int(v) for v in g['shape']]
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(int(g['x_seed']))) * 2 - 1
    m._classification_threshold = float(g['results_thr'])
    m._nms_cfg = dict(type='nms', iou_thr=float(g['results_iou']))
    with torch.no_grad():
        out = m(x.cuda())
    res = m.get_results(out, [dict(resized_height=H, resized_width=W, resize_scale=1.0)] * N)
    ref = json.loads(str(g['results']))
    unmatched = 0
    total = 0
    for n in range(N):
        used = set()
        for r in ref[n]:
            total += 1
            best = max(((i, _iou(r[2:], q[2:])) for i, q in enumerate(res[n]) if i not in used and q[0] == r[0]),
                       key=lambda t: t[1], default=(None, 0.0))
            if best[0] is not None and best[1] >= _IOU and abs(res[n][best[0]][1] - r[1]) <= _DSCORE:
                used.add(best[0])
            else:
                unmatched += 1
        unmatched += len(res[n]) - len(used)
    print('%s %s: %d reference detections, %d unmatched' % (name, precision, total, unmatched))
    _record('end_to_end/%s/%s' % (name, precision), dict(reference_detections=total, unmatched=unmatched))
    assert unmatched <= _UNMATCHED_GATE[('end_to_end/' + name, precision)], (unmatched, total)


@pytest.mark.parametrize('precision', ['fp16', 'fp32_storage'])
@pytest.mark.parametrize('tag', ['predict_py', 'q90'])
def test_baseline_config1_predict_for_single_image_matches_the_reference(tag, precision):
    """BASELINE config 1: WIDERFACE_LFD_XS.predict_for_single_image on the seeded 640 x 480 uint8 frame against the
    rows the REAL reference's predict_for_single_image returned on its CPU path (tests/golden/make_golden_config1.py):
    detections matched by class, IoU >= 0.99 and score within 2e-3.  In 'fp16' mode candidates whose score sits
    within the forward tolerance of the threshold may differ -- counted and bounded; in 'fp32_storage' mode (lfd.py:544-655 at
    the reference's precision) the reference's 1148 / 314 rows come back row for row."""
    g = load_golden('ref_config1_predict.npz')
    img = np.random.default_rng(0).integers(0, 256, (480, 640, 3)).astype(np.uint8)

    def aug(sample):   # simple_normalize (augmentation_pipeline.py:31-36)
        sample['image'] = ((sample['image'].astype(np.float32) / 255 - 0.5) / 0.5)
        return sample
    m = configs.build_model('WIDERFACE_LFD_XS', seed=666)
    configs.perturb_weights(m, seed=1)
    m.precision = precision
    res = m.predict_for_single_image(img, aug, classification_threshold=float(g[tag + '/thr']), nms_threshold=float(g[tag + '/iou']))
    ref = json.loads(str(g[tag + '/results']))
    used, unmatched = set(), 0
    for r in ref:
        best = max(((i, _iou(r[2:], q[2:])) for i, q in enumerate(res) if i not in used and q[0] == r[0]),
                   key=lambda t: t[1], default=(None, 0.0))
        if best[0] is not None and best[1] >= _IOU and abs(res[best[0]][1] - r[1]) <= _DSCORE:
            used.add(best[0])
        else:
            unmatched += 1
    unmatched += len(res) - len(used)
    print('config 1 %s %s: %d reference detections, %d unmatched' % (tag, precision, len(ref), unmatched))
    rec = dict(reference_detections=len(ref), unmatched=unmatched)
    if precision == 'fp32_storage':
        # IDENTITY, not only matching: the rows come back in the reference's ORDER -- row i of ours is row i
        # of the reference's (same class, IoU >= 0.99, score within 2e-3).  multiclass_nms orders by score (nms.py:161-220): two
        # detections whose scores differ by less than the forward's 2e-6 could swap; the count is recorded and gated at what
        # was measured on the MI355X (0 for both thresholds).
        assert len(res) == len(ref), (len(res), len(ref))
        out_of_order = sum(0 if (q[0] == r[0] and _iou(r[2:], q[2:]) >= _IOU and abs(q[1] - r[1]) <= _DSCORE) else 1 for r, q in zip(ref, res))
        rec['rows_out_of_order'] = out_of_order
        print('   rows not equal to the reference row of the same position: %d' % out_of_order)
    _record('config1/%s/%s' % (tag, precision), rec)
    assert unmatched <= _UNMATCHED_GATE[('config1/' + tag, precision)], (unmatched, len(ref))
    if precision == 'fp32_storage':
        assert rec['rows_out_of_order'] <= _ORDER_GATE['config1/' + tag], rec


def test_predict_for_single_image_api():
    m = configs.build_model('WIDERFACE_LFD_XS')
    configs.perturb_weights(m)
    img = np.random.default_rng(0).integers(0, 256, (120, 160, 3)).astype(np.uint8)

    def aug(sample):   # simple_normalize (augmentation_pipeline.py:31-36)
        sample['image'] = ((sample['image'].astype(np.float32) / 255 - 0.5) / 0.5)
        return sample
    with torch.no_grad():
        cls, _ = m.cuda().eval().forward_resident(torch.from_numpy(aug({'image': img})['image'][None].transpose(0, 3, 1, 2)).cuda())
        thr = float(np.quantile(cls.sigmoid().cpu().numpy(), 0.9))
    res = m.predict_for_single_image(img, aug, classification_threshold=thr, nms_threshold=0.3, class_agnostic=True)
    assert isinstance(res, list) and len(res) > 0 and all(len(r) == 6 and isinstance(r[0], int) for r in res)
    assert m._nms_cfg == dict(type='nms', iou_thr=0.3, class_agnostic=True)     # sticky mutation (lfd.py:630-633)
    scores = [r[1] for r in res]
    assert scores == sorted(scores, reverse=True) and min(scores) > thr
    assert all(0 <= r[2] <= 160 and 0 <= r[3] <= 120 for r in res)
    assert m.predict_for_single_image(img, aug, classification_threshold=0.9999999) == []


def test_train_step_runs_and_decreases_loss():
    """Training outer-loop contract (executor.py:191-211 + optimizer_hook.py:26-36): forward ->
    get_loss -> backward -> clip -> SGD step, with the HIP focal / IoU kernels in the loss."""
    m = configs.build_model('WIDERFACE_LFD_XS').cuda().train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(0)
    x = torch.rand(2, 3, 128, 128).cuda() * 2 - 1
    ann = [(np.array([[20, 24, 40, 44], [70, 60, 30, 36]], np.float32), np.zeros(2, np.int64)),
           (np.array([[50, 50, 16, 18]], np.float32), np.zeros(1, np.int64))]
    losses = []
    for _ in range(6):
        out = m(x)
        lo = m.get_loss(out, ann)
        opt.zero_grad()
        lo['loss'].backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10, norm_type=2)
        opt.step()
        losses.append(lo['loss_values']['loss'])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    del rng


def test_bench_prints_one_json_line_with_the_contract_fields():
    """bench.py (the driver's entry point): the full JSON line with the contract's keys, the roofline object of the dominant
    kernel and -- at N = 1 -- the CPU baseline and the bs-1 latency objects; then, LAST, the same contract line compacted to
    < 2 KB (the driver keeps a 2 KB tail: headline keys, roofline, cpu_baseline, serial / HIP-event figures, summaries)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1'],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 2, lines
    d = json.loads(lines[0])
    c2 = json.loads(lines[1])
    assert len(lines[1]) < 2048
    for k in ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'precision_mode'):
        assert c2[k] == d[k], k
    assert "fp32_storage" in c2['metric'] and c2['metric'].startswith(d['metric'][:40])      # (the compact line shortens the sentence)
    assert c2['roofline']['frac'] == d['roofline']['frac'] and c2['cpu_baseline']['value'] == d['cpu_baseline']['value']
    assert c2['ms_per_step_serial'] == d['ms_per_step_serial'] and c2['step_ms_hip_events']['median'] > 0
    assert c2['train']['ms_per_iter'] == d['train']['ms_per_iter']
    # round 6: `value` is quoted in the tolerance-compliant mode; the faster 'fp16' mode and the sustained (>= 1 s, gap-free)
    # rates ride along; BASELINE configs 3 / 4 and the 640 x 480 frames are timed in the headline mode too
    assert d['precision_mode'] == 'fp32_storage' and d['precise']['images_per_s_bs8'] == d['value']
    assert d['fp16_mode']['images_per_s'] > d['value'] and c2['fp16_mode']['images_per_s'] == d['fp16_mode']['images_per_s']
    for mode in (d, d['fp16_mode']):
        su = mode['images_per_s_sustained']
        assert su['pipelined']['seconds'] >= 0.5 and su['pipelined']['images_per_s'] > 0 and su['serial']['images_per_s'] > 0
    assert c2['images_per_s_sustained']['pipelined'] == d['images_per_s_sustained']['pipelined']['images_per_s']
    for k in ('config3', 'config4'):
        assert d['configs'][k]['precision_mode'] == 'fp32_storage' and d['configs'][k]['ms_per_step'] > d['configs'][k]['fp16']['ms_per_step'] > 0
        assert c2['configs'][k] == d['configs'][k]['ms_per_step']
    assert d['roofline']['frac_mfma_issued'] > d['roofline']['frac'] and d['fp16_mode']['roofline_conv3x3_s1_64']['frac'] > 0
    assert 'workload' in c2['config'] and 'model' not in c2['config']
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'latency_bs1'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f16' and d['data'] == 'synthetic'
    assert d['value'] > 0 and abs(d['value'] - 8 / d['ms_per_step'] * 1e3) < 0.01 * d['value']
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 2e-3 and 'traffic' in r
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['value'] > 0 and c['cores'] >= 1 and c['sample']
    assert d['latency_bs1']['forward_ms']['p50'] > 0 and d['latency_bs1']['end_to_end_ms']['p50'] > 0
    # extra keys the driver records with the line: the training iteration and the sibling meta-architectures
    assert d['train']['ms_per_iter'] > 0
    sib = d['siblings']
    assert isinstance(sib, list) and {r['config'] for r in sib} >= {'FCOS_FPN', 'LFDV2_SFPN', 'LFDV2_SIMPLE'}
    assert all(r['forward_ms'] > 0 and r['detect_ms'] > 0 and r['images_per_s'] > 0 for r in sib)
    # round 4: SURVEY 8d's remaining points (K = 4096 / IoU 0.3 at the headline shape, 640x480 frames), the all-backbone roofline
    # entry, and which precision mode meets which gate
    cf = d['configs']
    assert cf['stress_k4096_iou03']['ms_per_step'] > 0 and 3000 < cf['stress_k4096_iou03']['candidates_per_image'] < 5000
    assert cf['stress_k4096_iou03']['overflow'] == 0 and cf['stress_k4096_iou03']['iou_thr'] == 0.3
    assert cf['frames_640x480']['bs8']['ms_per_step'] > 0 and cf['frames_640x480']['bs1']['points_per_image'] == 6460
    rb = d['roofline_backbone_3x3']
    assert rb['bound'] == 'mfma' and 0 < rb['frac'] < 1 and abs(rb['frac'] - rb['achieved'] / rb['peak']) < 2e-3
    assert "'fp16'" in d['metric'] and 'fp32_storage' in d['metric'] and len(d['parity_gates']) == 2
    assert d['fp16_mode']['latency_bs1']['end_to_end_ms']['p50'] > 0 and d['fp16_mode']['configs']['frames_640x480']['bs8']['ms_per_step'] > 0


def test_two_batches_in_flight_on_two_streams_match_the_serial_step():
    """LFD.detect_resident(slot=...) -- independent activation / output / workspace buffers per slot -- lets bench.py keep two
    batches in flight on two HIP streams.  Different frames per slot, many overlapping replays: every slot's detections must
    equal what the same frames give when replayed alone."""
    m = configs.build_model('WIDERFACE_LFD_S')
    configs.perturb_weights(m)
    m.eval().cuda()
    m.use_graph = True
    g = torch.Generator(device='cuda').manual_seed(3)
    xs = [(torch.rand(2, 270, 480, 3, device='cuda', generator=g) * 2 - 1).half() for _ in range(2)]
    meta = torch.tensor([[480.0, 270.0, 1.0]] * 2).cuda()
    with torch.no_grad():
        cls, _ = m.forward_resident(xs[0])
        thr = float(torch.quantile(cls.float().sigmoid().reshape(-1), 0.99))
        ref = []
        for sl in range(2):
            o = m.detect_resident(xs[sl], meta, score_thr=thr, slot=0)
            torch.cuda.synchronize()
            ref.append((o.counts.clone(), o.dets.clone(), o.labels.clone()))
        assert not torch.equal(ref[0][0], ref[1][0]) or not torch.equal(ref[0][1], ref[1][1])
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        outs = [None, None]
        for it in range(40):
            sl = it & 1
            with torch.cuda.stream(streams[sl]):
                outs[sl] = m.detect_resident(xs[sl], meta, score_thr=thr, slot=sl)
        torch.cuda.synchronize()
    for sl in range(2):
        assert torch.equal(outs[sl].counts, ref[sl][0])
        for n in range(2):
            k = int(ref[sl][0][n, 1])
            assert k > 0 and torch.equal(outs[sl].dets[n, :k], ref[sl][1][n, :k]) and torch.equal(outs[sl].labels[n, :k], ref[sl][2][n, :k])


@pytest.mark.parametrize('shape,q', [((2, 270, 480, 3), 0.95), ((1, 1080, 1920, 3), 0.995), ((3, 200, 312, 3), 0.9), ((2, 203, 317, 3), 0.9)])
def test_head_pass_that_appends_its_own_candidates_equals_the_unfused_step(shape, q):
    """lfd_head_forward_decode_f16 (the head's output pass thresholds, decodes and appends the candidates; sort + mask + scan
    follow through lfd_detect_from_candidates) against forward -> fp32 logits -> lfd_detect_batched: identical counts,
    detections, labels and point indices (the append order is arbitrary, the sort key (score, point) is not), over several
    replays of the captured graph (the candidate counters re-arm themselves)."""
    from lfd_amd import engine
    m = configs.build_model('WIDERFACE_LFD_S')
    configs.perturb_weights(m)
    m.eval().cuda()
    x = (torch.rand(*shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)) * 2 - 1).half()
    meta = torch.tensor([[float(shape[2]), float(shape[1]), 1.0]] * shape[0]).cuda()
    with torch.no_grad():
        cls, _ = m.forward_resident(x)
        thr = float(torch.quantile(cls.float().sigmoid().reshape(-1)[:4000000], q))
        m.use_graph = False
        u = m.detect_resident(x, meta, score_thr=thr, max_candidates=4096)      # eager: forward + lfd_detect_batched
        torch.cuda.synchronize()
        ref = [t.clone() for t in (u.counts, u.dets, u.labels, u.point)]
        assert int(ref[0][:, 2].max()) == 0 and int(ref[0][:, 0].min()) > 10
        m.use_graph = True
        plan = engine.get_plan(m, m._backbone, m._neck, m._head, x.device)
        for _ in range(3):
            f = m.detect_resident(x, meta, score_thr=thr, max_candidates=4096)
            torch.cuda.synchronize()
            assert torch.equal(f.counts, ref[0])
            for n in range(shape[0]):
                k = int(ref[0][n, 1])
                assert k > 0
                assert torch.equal(f.dets[n, :k], ref[1][n, :k]) and torch.equal(f.labels[n, :k], ref[2][n, :k])
                assert torch.equal(f.point[n, :k], ref[3][n, :k])
    # capacity overflow: the counters keep counting, the slots beyond the capacity are dropped, the flag is raised (the
    # retained subset is arbitrary in the fused path -- documented -- so only the bookkeeping is compared)
    with torch.no_grad():
        m.use_graph = False
        u = m.detect_resident(x, meta, score_thr=thr, max_candidates=8)
        torch.cuda.synchronize()
        uc = u.counts.clone()
        m.use_graph = True
        for _ in range(2):
            f = m.detect_resident(x, meta, score_thr=thr, max_candidates=8)
            torch.cuda.synchronize()
            assert torch.equal(f.counts[:, [0, 2, 3]], uc[:, [0, 2, 3]]) and int(f.counts[:, 2].min()) == 1
            assert int(f.counts[:, 1].max()) <= 8 and int(f.counts[:, 1].min()) >= 1
    # the fused pass really ran: this model / descriptor is covered
    desc, _ = m._detect_desc(thr, m._nms_cfg.get('iou_thr', 0.5), m._nms_cfg.get('class_agnostic', False), 4096)
    assert plan.decode_supported(plan.state_for(*([shape[0], shape[1], shape[2]]), 0), desc)
