"""bench.py -- BASELINE.json metric: images/sec, WIDERFACE_LFD_S end-to-end inference
(forward + per-location decode + score threshold + NMS, results left resident on the device),
synthetic 1920x1080 frames, batch 8 per GPU, fp16 NHWC input already resident in HBM.

`value` is quoted in LFD.precision = 'fp32_storage' -- the precision mode whose cls / bbox tensors are within north_star's
1e-3 of the reference's fp32 path (raw logits <= 1e-4, tests/test_gpu_precise.py).  Rounds 1-5 quoted it in the 'fp16' mode
(sigma within 2.5e-3: outside that tolerance); that mode is still measured, by the same timed region, under `fp16_mode`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; images shard across ranks with NO data-path collective (weak scaling:
8 frames per GPU per step); rank 0 prints ONE JSON line.  A plain `python bench.py --gpus N` (no launcher in the
environment) re-executes itself under torch.distributed.run with N ranks; `ranks_seen` in the line is an RCCL all-reduce
of ones, i.e. the number of ranks that really took part.  `roofline` = the dominant kernel class
timed live with HIP events on the launch stream; `cpu_baseline` = the oracle's port of the
reference CPU path (PyTorch fp32 eager NCHW forward + decode + greedy NMS) on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL = 'WIDERFACE_LFD_S'
BATCH, H, W = 8, 1080, 1920
NBUF = 4                  # distinct resident frame buffers rotated through the steps: 4 x 99.5 MB > the 256 MiB Infinity Cache,
                          # so the stem's reads are cold HBM reads (one buffer replayed every step would sit in the cache)
TARGET_K = 256            # candidates per image (SURVEY 8d "sparse-realistic" load), IoU 0.4
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
HEADLINE_MODE = 'fp32_storage'   # the precision mode `value` is quoted in: the one inside north_star's 1e-3 (round 6)
PMC_SOURCE_PRECISE = ('profiles/pmc_traffic_precise.json: HBM bytes per launch of the fp32_storage kernels (FETCH_SIZE x 2 + WRITE_SIZE, separate '
                      'rocprofv3 --pmc passes, tools/collect_profiles_p2.sh) committed with the kernels -- not re-measured on this box')
PMC_SOURCE = ('profiles/pmc_traffic.json: HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, '
              'tools/collect_profiles.sh) committed with the kernels -- a constant of the code, not re-measured on this box')


def conv_flops(n, oh, ow, cin, cout, ks):
    return 2.0 * n * oh * ow * cin * cout * ks * ks


def kernel_breakdown(model, plan, st, x, fmt, reps=20, timer=None):
    """Times every launch of the plan with HIP events on the launch stream (torch's current
    stream is the stream the C ABI receives).  Returns {class: dict(time_us, launches, flops, bytes)}.
    `timer(fn) -> us` replaces the HIP-event loop (tools/power_trace_r6.py: seconds of back-to-back launches under rocm-smi)."""
    import ctypes as C
    from lfd_amd import _lib, engine, ops
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    l = lib()
    z = ops.zero_line(plan.device)
    classes = {}

    def timed(fn, label=''):
        if timer is not None:
            return timer(fn, label)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps   # us

    def add(name, us, flops, nbytes):
        c = classes.setdefault(name, dict(time_us=0.0, launches=0, flops=0.0, bytes=0.0))
        c['time_us'] += us
        c['launches'] += 1
        c['flops'] += flops
        c['bytes'] += nbytes

    n = st.n
    if plan.stem_fused is not None:
        c0, w1, b1, w2, b2, w3, b3, w4, b4, dst = plan.stem_fused
        so = st.bufs[dst]
        h1, w1_ = (st.h + 1) // 2, (st.w + 1) // 2
        us = timed(lambda: check(l.lfd_stem_faster_fused_f16(ptr(x), fmt, n, st.h, st.w, c0, ptr(w1), ptr(b1), ptr(w2),
                                                             ptr(b2), ptr(w3), ptr(b3), ptr(w4), ptr(b4), ptr(so),
                                                             stream_ptr()), 'stem'), 'k_stem2x 8x1080p')
        fl = (conv_flops(n, h1, w1_, 3, c0, 3) + conv_flops(n, h1, w1_, c0, c0, 1) +
              conv_flops(n, so.shape[1], so.shape[2], c0, c0, 3) + conv_flops(n, so.shape[1], so.shape[2], c0, c0, 1))
        add('whole faster-stem fused: 3x3s2+1x1+3x3s2+1x1 (k_stem2x)', us, fl, x.numel() * x.element_size() + so.numel() * 2)
    else:
        c0, w1, b1, w2, b2 = plan.stem_first
        so = st.bufs[plan.stem_out]
        us = timed(lambda: check(l.lfd_stem_conv_f16(ptr(x), fmt, n, st.h, st.w, c0, ptr(w1), ptr(b1), ptr(w2), ptr(b2),
                                                     ptr(so), stream_ptr()), 'stem'))
        add('stem_3x3s2_3toC+1x1 (k_stem)', us, conv_flops(n, so.shape[1], so.shape[2], 3, c0, 3) +
            (conv_flops(n, so.shape[1], so.shape[2], c0, c0, 1) if w2 is not None else 0), x.numel() * x.element_size() + so.numel() * 2)
    skip = -1
    for ci, c in enumerate(plan.convs):
        if ci == skip:
            continue
        src, dst = st.bufs[c.src], st.bufs[c.dst]
        if c.down is not None and engine._use_fused_down(n, src.shape[1], src.shape[2]):
            # the stage's first block in one launch (csrc/down.hip): 3x3 s2 + its 1x1 s2 branch + 3x3 s1 + add; algorithmic
            # bytes = the input map read once + the output written once (y1 and the branch never reach HBM)
            c2 = plan.convs[c.down]
            dst2 = st.bufs[c2.dst]
            us = timed(lambda: check(l.lfd_downblock_fused_f16(n, src.shape[1], src.shape[2], ptr(src), ptr(dst2), ptr(c.w), ptr(c.b),
                                                               ptr(c.ds[0]), ptr(c.ds[1]), ptr(c2.w), ptr(c2.b), ptr(z), stream_ptr()),
                                     'downblock'), 'k_down64 -> %dx%d' % (dst2.shape[1], dst2.shape[2]))
            fl = (conv_flops(n, dst2.shape[1], dst2.shape[2], 64, 64, 3) * 2 + conv_flops(n, dst2.shape[1], dst2.shape[2], 64, 64, 1))
            add('downblock_fused_conv3x3_s2+1x1_s2+conv3x3_s1_64to64 (k_down64)', us, fl, (src.numel() + dst2.numel()) * 2)
            skip = c.down
            continue
        if c.blk128 is not None and engine._use_fused_block128(n, src.shape[1], src.shape[2]):
            # a 128-channel block of the last stage in one launch (csrc/block128.hip); algorithmic bytes = the map read once +
            # written once + both filters once
            c2 = plan.convs[c.blk128]
            dst2 = st.bufs[c2.dst]
            us = timed(lambda: check(l.lfd_fasterblock128_fused_f16(n, src.shape[1], src.shape[2], ptr(src), ptr(dst2), ptr(c.w), ptr(c.b),
                                                                    ptr(c2.w), ptr(c2.b), stream_ptr()), 'block128'), 'k_block128 %dx%d' % (dst2.shape[1], dst2.shape[2]))
            add('fasterblock128_fused_2x_conv3x3_s1_128to128 small map (k_block128)', us,
                2 * conv_flops(n, dst2.shape[1], dst2.shape[2], 128, 128, 3), (src.numel() + dst2.numel()) * 2 + 2 * 128 * 128 * 9 * 2)
            skip = c.blk128
            continue
        d = _lib.ConvDesc(n, src.shape[1], src.shape[2], c.cin, c.cout, c.ks, c.stride, int(c.relu),
                          c.cout if c.tail else 0, 1 if c.tail else 0)
        if c.blk is not None:
            fn = lambda: check(l.lfd_fasterblock_fused_f16(n, src.shape[1], src.shape[2], ptr(src), ptr(dst), ptr(c.w), ptr(c.b),  # noqa: E731
                                                           ptr(c.blk[0]), ptr(c.blk[1]), ptr(z), stream_ptr()), 'block')
        elif c.ds is not None:
            fn = lambda: check(l.lfd_conv2d_downsample_nhwc_f16(C.byref(d), ptr(src), ptr(dst), ptr(c.w), ptr(c.b),  # noqa: E731
                                                                ptr(c.ds[0]), ptr(c.ds[1]), ptr(st.bufs[c.ds[2]]), ptr(z),
                                                                stream_ptr()), 'conv+ds')
        else:
            fn = lambda: check(l.lfd_conv2d_nhwc_f16(C.byref(d), ptr(src), ptr(dst), ptr(c.w), ptr(c.b),  # noqa: E731
                                                     ptr(st.bufs[c.res]) if c.res is not None else None,
                                                     ptr(c.tail[0]) if c.tail else None, ptr(c.tail[1]) if c.tail else None,
                                                     ptr(z), stream_ptr()), 'conv')
        us = timed(fn, '%s %d->%d k%d s%d -> %dx%d' % ('k_block64(_rows)' if c.blk is not None else 'k_conv', c.cin, c.cout, c.ks, c.stride, dst.shape[1], dst.shape[2]))
        fl = conv_flops(n, dst.shape[1], dst.shape[2], c.cin, c.cout, c.ks)
        if c.ds is not None:
            fl += conv_flops(n, dst.shape[1], dst.shape[2], c.cin, c.cout, 1)
        if c.tail:
            fl += conv_flops(n, dst.shape[1], dst.shape[2], c.cout, c.cout, 1)
        by = (src.numel() + dst.numel() + (dst.numel() if c.res is not None else 0)) * 2
        if c.ks == 1 and c.stride == 2:
            by = (src.numel() // 4 + dst.numel()) * 2
        if c.ds is not None:
            by += dst.numel() * 2
        name = 'conv%dx%d_s%d_%dto%d%s (k_conv)' % (c.ks, c.ks, c.stride, c.cin, c.cout,
                                                   '+1x1' if c.tail else ('+downsample1x1s2' if c.ds is not None else ''))
        if c.cin == 128 and c.cout == 128 and c.ks == 3 and c.stride == 1 and not c.tail and n * dst.shape[1] * dst.shape[2] <= 16384:
            name = 'conv3x3_s1_128to128 small map, split-K (k_conv128_splitk)'
        if c.blk is not None:
            # whole residual block: two 3x3 convs; algorithmic bytes = the map read once + written once (the identity is
            # the same tensor as the input; the intermediate never reaches HBM)
            fl *= 2
            by = (src.numel() + dst.numel()) * 2
            name = 'fasterblock_fused_2x_conv3x3_s1_64to64 (k_block64_rows on large maps, k_block64 on small ones)'
        add(name, us, fl, by)
    us = timed(lambda: plan.run_head(st), 'neck+head: k_head2 x3 + k_gn_finalize x2')
    hf = 0.0
    hb = 0.0
    for lv, (hh, ww) in zip(plan.levels, st.sizes):
        px = n * hh * ww
        per_tower = 3 * 2 * px * lv.cin * 128 + (3 + 2) * 2 * px * 128 * 128 + 2 * px * 128 * 32
        hf += per_tower * len(lv.towers)
        # bytes a pass structure of this kind has to move: the tap read once per pass, the outputs, AND the deliberate tower-1
        # hand-off (pass 2 writes conv2's operands, 256 B per pixel, the output pass reads them: DESIGN lesson 19) -- VERDICT r2 #9
        hb += (3 * px * lv.cin * 2 + 2 * px * 256) * len(lv.towers) + px * (plan.cls_channels + 4) * 4
    add('neck+head 3-pass GN recompute (k_head2 x3 + k_gn_finalize x2)', us, hf, hb)
    return classes


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(reps=5):
    """The reference's CPU path on THIS host, bounded sample: 1 warm-up + `reps` timed 1920x1080 frames (bs 1), median.
    Forward = oracle/net_oracle.py (the reference's modules restated functionally: PyTorch fp32 eager NCHW conv /
    BatchNorm / GroupNorm -- `kind: port`); post-processing = sigmoid, decode, strict threshold (torch ops as in
    lfd.py:449-499) + greedy NMS through the REFERENCE's own compiled nms_cpu.cpp (oracle/_ref, nms_cpu.cpp:7-66) when the
    prebuilt module is present (`nms_leg: reference`), else the C port.  Reported next to the GPU number; not a target."""
    from oracle import build_ref, net_oracle
    from lfd_amd import configs
    arch = configs.ARCHS[MODEL]
    m = configs.build_model(MODEL)
    configs.perturb_weights(m)
    m.eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    host = os.cpu_count() or 1
    # torch's CPU conv does not scale past a few tens of threads on this class of host (a 256-thread run of one 1080p
    # frame measured > 40 s, 16 threads fastest on the MI355X box's host): threads used are stated as `cores`
    cores = min(host, int(os.environ.get('LFD_CPU_BASELINE_THREADS', '16')))
    torch.set_num_threads(cores)
    ref_ext = None
    try:
        ref_ext = build_ref.load_ref()
    except Exception:
        ref_ext = None
    x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(0)) * 2 - 1
    strides = net_oracle.strides_of(arch)
    t_fwd, t_all, kept = [], [], 0
    with torch.no_grad():
        for i in range(reps + 1):
            t0 = time.perf_counter()
            cls, reg, sizes = net_oracle.lfd_forward(sd, arch, x)
            t1 = time.perf_counter()
            if i == 0:
                thr = float(np.partition(cls[0].sigmoid().numpy().reshape(-1), -TARGET_K)[-TARGET_K])
                t1 = time.perf_counter()
            if ref_ext is not None:
                sc = cls[0].sigmoid()
                bx = torch.from_numpy(net_oracle.decode_boxes(reg[0], sizes, strides, arch['regression_ranges'], 'sigmoid',
                                                              'union', (H, W), 1.0))
                idx = torch.nonzero(sc[:, 0] > thr)[:, 0]
                dets = torch.cat([bx[idx], sc[idx]], 1).contiguous()
                keep = ref_ext.nms(dets, 0.4)                 # the reference's nms_cpu_kernel, one class -> offsets are 0
                kept = int(keep.numel())
            else:
                d, l, _, _ = net_oracle.get_results_single(cls[0].numpy(), reg[0].numpy(), sizes, strides, arch, thr, 0.4, False,
                                                           (H, W), 1.0)
                kept = len(l)
            t2 = time.perf_counter()
            if i:
                t_fwd.append(t1 - t0)
                t_all.append(t2 - t0)
    tf, ta = float(np.median(t_fwd)), float(np.median(t_all))
    return dict(value=round(1.0 / ta, 3), unit='images/s', cores=cores, kind='port',
                forward_only=round(1.0 / tf, 3), nms_leg='reference' if ref_ext is not None else 'port',
                host_cores=host, cpu_model=_cpu_model(), reps=reps, kept=kept,
                sample='1 warm-up + %d timed 1920x1080 frames (bs 1), median; value = forward + sigmoid + decode + threshold + '
                       'greedy NMS at K=%d candidates, forward_only = network alone; torch %s fp32 eager NCHW on %d of the '
                       'host\'s %d hardware threads' % (reps, TARGET_K, torch.__version__, cores, host))


def train_bench(dev, steps=10, warmup=3, world=1, rank=0):
    """BASELINE.json configs[4]: WIDERFACE_LFD_S train-from-scratch iteration, synthetic 640x640, bs 32 PER GPU --
    forward + device targets + fused get_loss + hand-written backward + clip_grad_norm_ + SGD (lfd_amd.train.train_step).
    826 GFLOP per GPU and iteration = 3 x the 8.606 GFLOP/img forward x 32.  At world > 1 (bench.py --gpus N,
    one process per GPU) every rank runs it: image-parallel DDP -- the all-reduced loss normalisers and ONE all-reduce of
    the flat gradient buffer per iteration over RCCL -- eagerly (train_step) and as three HIP graphs with the two collectives
    between them (GraphedTrainStep under torch.distributed); per-iteration time = MAX over ranks of the rank medians.
    Extra key of the bench line so that the driver records it; the headline `value` is the inference metric."""
    from lfd_amd import configs, optim, train
    torch.manual_seed(0)
    m = configs.build_model(MODEL).to(dev).train()
    opt = optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    rng = np.random.default_rng(0)
    bs, size = 32, 640
    x = torch.randn(bs, 3, size, size, device=dev)
    ann = []
    for _ in range(bs):
        wh = np.exp(rng.uniform(np.log(8), np.log(200), (6, 2)))
        xy = rng.uniform(0, 1, (6, 2)) * (np.array([size, size]) - wh).clip(1)
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), np.zeros(6, np.int64)))
    clip = dict(max_norm=10, norm_type=2)

    def timed(fn):
        for _ in range(warmup):
            lv, _ = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            lv, _ = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return ts, lv

    ts_eager, lv = timed(lambda: train.train_step(m, opt, x, ann, clip, True))
    # the same iteration replayed as one HIP graph (lfd_amd.train.GraphedTrainStep: ~300 launches, one host call; three graphs
    # with the two RCCL all-reduces between them when a process group is live)
    graphed = None
    ts_b2b = 0.0
    step = None
    try:
        step = train.GraphedTrainStep(m, opt, clip, max_boxes=1024)
    except Exception as e:       # keep the eager number
        graphed = repr(e)
    if dist.is_initialized():
        # the graphed iterations contain collectives: either every rank runs them or none does (a rank that fell back to
        # eager alone would leave the others blocked in an all-reduce)
        flag = torch.tensor([0 if step is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            step = None
            graphed = graphed or 'another rank could not build the graphed step'
    ts = ts_eager
    if step is not None:
        ts, lv = timed(lambda: step(x if step.x is None else step.x, ann, True))     # frames written into the step's own buffer
        # the same iterations enqueued back to back (sync=False: no read-back between replays), the last loss read at the end.
        # Measured round 4: NOT faster (7.23 against 7.08 ms) -- without the ~0.13 ms host gap between replays the chip sits
        # at its power limit and clocks lower; reported, not used as the headline of this key
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pend, _ = step(step.x, ann, True, sync=False)
        pend.get()
        torch.cuda.synchronize()
        ts_b2b = (time.perf_counter() - t0) / steps
        graphed = True
    dt, dt_eager = float(np.median(ts)), float(np.median(ts_eager))
    per_rank = [round(dt * 1e3, 3)]
    if dist.is_initialized():
        t = torch.tensor([dt, dt_eager, ts_b2b], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [round(float(a[0]) * 1e3, 3) for a in allt]
        dt, dt_eager, ts_b2b = (max(float(a[i]) for a in allt) for i in range(3))
    gf = 3 * 8.606 * bs
    return dict(workload='WIDERFACE_LFD_S train step 640x640 bs 32 per GPU (forward + targets + loss + backward + clip + SGD), fp16 '
                         'activations / fp32 accumulate + parameters' + ('; image-parallel DDP over %d ranks: all-reduced loss '
                         'normalisers + one all-reduce of the flat gradient buffer per iteration (RCCL)' % world if dist.is_initialized() else ''),
                n_gpus=world, ranks_seen=len(per_rank), ms_per_iter_per_rank=per_rank,
                ms_per_iter=round(dt * 1e3, 3), images_per_s=round(world * bs / dt, 1), gflop_per_iter_per_gpu=round(gf, 1),
                tflops_per_gpu=round(gf / dt / 1e3, 1), frac_mfma=round(gf / dt / 1e3 / MFMA_PEAK_TFLOPS, 4), steps=steps, warmup=warmup,
                loss=lv['loss'], hip_graph=graphed, graphs_per_iter=(3 if dist.is_initialized() else 1),
                ms_per_iter_eager=round(dt_eager * 1e3, 3),
                ms_per_iter_back_to_back=(round(ts_b2b * 1e3, 3) if graphed is True else None),
                note='ms_per_iter: the iteration as HIP graph(s), the loss values read back after every iteration (median; MAX '
                     'over ranks); ms_per_iter_back_to_back: iterations enqueued without read-back (GraphedTrainStep sync=False); '
                     'ms_per_iter_eager: ~300 eager launches per iteration')


def siblings_bench(dev, reps=20, n=8, h=720, w=1280):
    """SURVEY 8(f) rank 4 on record: the sibling meta-architectures (FCOS / LFDv2 over FPN / SimpleFPN necks, 3x3 heads;
    lfd_amd.configs.SIBLINGS) -- forward as one HIP graph + lfd_detect_batched_ex, HIP events, median."""
    from lfd_amd import configs
    out = []
    for name in sorted(configs.SIBLINGS):
        spec = configs.SIBLINGS[name]
        model = configs.build_sibling_model(name, seed=1).eval().to(dev)
        model.use_graph = True
        x = (torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)
        meta = torch.tensor([[float(w), float(h), 1.0]] * n, dtype=torch.float32, device=dev)
        fwd = (lambda: model.forward_resident(x)) if hasattr(model, 'forward_resident') else (lambda: model(x))
        with torch.no_grad():
            outs = fwd()
            if len(outs) == 3:
                sc = outs[0].sigmoid() * outs[2].sigmoid()
            elif spec.get('classification_loss_type') == 'CrossEntropyLoss':
                sc = outs[0].softmax(-1)[..., :-1]
            else:
                sc = outs[0].sigmoid()
            model._classification_threshold = float(torch.quantile(sc.flatten()[:4_000_000].float(), 0.98))
            torch.cuda.synchronize()
            ts = {}
            for key, fn in (('forward_ms', fwd), ('detect_ms', lambda: model.detect(outs, meta))):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                v = []
                for _ in range(reps):
                    s.record()
                    fn()
                    e.record()
                    e.synchronize()
                    v.append(s.elapsed_time(e))
                ts[key] = round(float(np.median(v)), 4)
            counts = model.detect(outs, meta).counts.cpu()
        out.append(dict(config=name, meta_arch=spec['meta'], batch=n, input=[h, w], points_per_image=int(outs[0].shape[1]),
                        images_per_s=round(n / ((ts['forward_ms'] + ts['detect_ms']) * 1e-3), 1),
                        kept_per_image=float(counts[:, 1].float().mean()), **ts))
    return out


def _event_median_ms(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    v = []
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(reps):
        s.record()
        fn()
        e.record()
        e.synchronize()
        v.append(s.elapsed_time(e))
    return float(np.median(v))


def other_configs_bench(dev):
    """BASELINE.json configs[2] and configs[3] on ONE GPU, extra key `configs` (the headline stays configs[1]):
      config3  WIDERFACE_LFD_L, one 3840x2160 frame                     (large-activation path)
      config4  TT100K_LFD_L, 4 x 1280x720 = one GPU's share of bs 32 / 8 GPUs, 45 classes, softmax scores, per-class NMS
    Each in BOTH precision modes (round 6: the top-level figures are the headline mode's, 'fp32_storage'; the 'fp16' mode's under
    `fp16`): the whole step (forward + decode + NMS) as one HIP graph, HIP-event median; network-only forward; the dominant
    kernel class with its roofline fraction (same live per-launch timing as the headline's `kernels`)."""
    from lfd_amd import configs, engine
    out = {}
    for key, name, (n, h, w), reps, k_cand in (('config3', 'WIDERFACE_LFD_L', (1, 2160, 3840), 20, 1024),
                                               ('config4', 'TT100K_LFD_L', (4, 720, 1280), 30, 256)):
        m = configs.build_model(name)
        configs.perturb_weights(m)
        m.eval().to(dev)
        m.max_candidates = 8192
        arch = configs.ARCHS[name]
        ce = arch['classification_loss_type'] == 'CrossEntropyLoss'
        gen = torch.Generator(device=dev).manual_seed(3)
        x = (torch.rand(n, h, w, 3, device=dev, generator=gen) * 2 - 1).half()
        meta = torch.tensor([[float(w), float(h), 1.0]] * n, dtype=torch.float32, device=dev)
        m._nms_cfg = dict(type='nms', iou_thr=0.4 if not ce else 0.1)
        res = {}
        for mode in (HEADLINE_MODE, 'fp16'):
            m.precision = mode
            m.use_graph = False
            cls, reg = m.forward_resident(x)
            sc = (cls[0].float().softmax(-1)[:, :-1] if ce else cls[0].float().sigmoid()).max(-1).values
            m._classification_threshold = float(torch.quantile(sc[:4_000_000], 1.0 - k_cand / sc.numel()))
            fwd_ms = _event_median_ms(lambda: m.forward_resident(x), 5, warm=1)
            m.use_graph = True
            step_ms = _event_median_ms(lambda: m.detect_resident(x, meta), reps)
            counts = m.detect_resident(x, meta).counts.cpu().numpy()
            d = dict(ms_per_step=round(step_ms, 4), images_per_s=round(n / step_ms * 1e3, 1), forward_eager_ms=round(fwd_ms, 4),
                     points_per_image=int(cls.shape[1]), candidates_per_image=float(counts[:, 0].mean()),
                     kept_per_image=float(counts[:, 1].mean()), overflow=int(counts[:, 2].max()))
            if mode == 'fp16':
                fmt, n_, h_, w_ = engine._input_format(x)
                plan = engine.get_plan(m, m._backbone, m._neck, m._head, dev)
                br = kernel_breakdown(m, plan, plan.state_for(n_, h_, w_), x, fmt, reps=5)
                tot = sum(c['time_us'] for c in br.values())
                nm, c = max(br.items(), key=lambda kv: kv[1]['time_us'])
                tf, gb = c['flops'] / c['time_us'] / 1e6, c['bytes'] / c['time_us'] / 1e3
                gflop = sum(c_['flops'] for c_ in br.values()) / 1e9
                d.update(kernel_sum_us=round(tot, 1), gflop_per_step=round(gflop, 1), frac_mfma_whole_step=round(gflop / step_ms / MFMA_PEAK_TFLOPS, 4),
                         roofline=dict(kernel=nm, bound='mfma' if tf / MFMA_PEAK_TFLOPS >= gb / HBM_PEAK_GBS else 'hbm',
                                       tflops=round(tf, 1), gbs=round(gb, 0), frac_mfma=round(tf / MFMA_PEAK_TFLOPS, 3),
                                       frac_hbm=round(gb / HBM_PEAK_GBS, 3), share_of_forward=round(c['time_us'] / tot, 3),
                                       avg_launch_us=round(c['time_us'] / c['launches'], 2)))
                del plan, br
            else:
                try:
                    rows = precise_breakdown(m, x, dev, reps=5)
                except Exception as e:      # (a layer shape without a plane kernel: the fp32-tensor plan has no per-class table)
                    rows = None
                    d['kernels_error'] = repr(e)
                if rows:
                    tot = sum(r['time_us_per_forward'] for r in rows)
                    r0 = rows[0]
                    d.update(kernel_sum_us=round(tot, 1),
                             head_share_of_forward=round(sum(r['time_us_per_forward'] for r in rows if r['kernel'].startswith('neck + head')) / tot, 3),
                             roofline=dict(kernel=r0['kernel'], bound=r0['bound'], achieved=r0['achieved'], peak=r0['peak'], unit=r0['unit'],
                                           frac=r0['frac'], frac_mfma_issued=r0['frac_mfma_issued'], frac_hbm=r0['frac_hbm'],
                                           share_of_forward=round(r0['time_us_per_forward'] / tot, 3), avg_launch_us=r0['avg_launch_us']))
            res[mode] = d
            del cls, reg
        out[key] = dict(workload='%s %d x %dx%d fp16 NHWC frames resident, forward + decode + NMS (one HIP graph)' % (name, n, w, h),
                        precision_mode=HEADLINE_MODE, **res[HEADLINE_MODE])
        out[key]['fp16'] = res['fp16']
        del m, x
        torch.cuda.empty_cache()
    return out


def stress_and_small_frame_bench(model, x, meta, dev, cls):
    """SURVEY 8d's remaining measurement points, extra keys under `configs` (the headline stays K = 256 / IoU 0.4 at 1080p):
      stress_k4096_iou03   the headline batch with the score threshold at the quantile that leaves ~4096 candidates per image
                           and IoU 0.3 (the NMS stress load: 4096 x 64-word suppression masks per image)
      frames_640x480       the 640x480 frame north_star lists beside 1920x1080: a batch of 8 and a single frame
    Each: the whole step (forward + decode + NMS) as one HIP graph, HIP-event median."""
    out = {}
    keep = (model._classification_threshold, dict(model._nms_cfg), model.use_graph)
    try:
        model.use_graph = True
        k = 4096
        thr = float(torch.quantile(cls.float().sigmoid().reshape(x.size(0), -1)[0], 1.0 - k / cls.shape[1]))
        model._classification_threshold, model._nms_cfg = thr, dict(type='nms', iou_thr=0.3)
        ms = _event_median_ms(lambda: model.detect_resident(x, meta), 50)
        counts = model.detect_resident(x, meta).counts.cpu().numpy()
        out['stress_k4096_iou03'] = dict(workload='WIDERFACE_LFD_S 8 x 1920x1080, ~4096 candidates per image, IoU 0.3, forward + decode + '
                                                  'NMS (one HIP graph, one batch in flight)', ms_per_step=round(ms, 4),
                                         images_per_s=round(x.size(0) / ms * 1e3, 1), score_thr=thr, iou_thr=0.3,
                                         candidates_per_image=float(counts[:, 0].mean()), kept_per_image=float(counts[:, 1].mean()),
                                         overflow=int(counts[:, 2].max()))
        model._classification_threshold, model._nms_cfg = keep[0], dict(keep[1])
        gen = torch.Generator(device=dev).manual_seed(17)
        small = {}
        for n in (8, 1):
            xs_ = (torch.rand(n, 480, 640, 3, device=dev, generator=gen) * 2 - 1).half()
            meta_ = torch.tensor([[640.0, 480.0, 1.0]] * n, dtype=torch.float32, device=dev)
            c_, _ = model.forward_resident(xs_)
            model._classification_threshold = float(torch.quantile(c_.float().sigmoid().reshape(n, -1)[0], 1.0 - 64.0 / c_.shape[1]))
            ms = _event_median_ms(lambda: model.detect_resident(xs_, meta_), 100)
            cn = model.detect_resident(xs_, meta_).counts.cpu().numpy()
            small['bs%d' % n] = dict(ms_per_step=round(ms, 4), images_per_s=round(n / ms * 1e3, 1), points_per_image=int(c_.shape[1]),
                                     candidates_per_image=float(cn[:, 0].mean()), kept_per_image=float(cn[:, 1].mean()))
        small['workload'] = 'WIDERFACE_LFD_S 640x480 fp16 NHWC resident, ~64 candidates per image, IoU 0.4, forward + decode + NMS (one HIP graph)'
        out['frames_640x480'] = small
        # the headline batch as the reference's data pipeline hands it over: uint8 NHWC, simple_normalize
        # (augmentation_pipeline.py:31-36) inside the fused stem
        x8 = torch.randint(0, 256, tuple(x.shape), device=dev, dtype=torch.uint8, generator=gen)
        c8, _ = model.forward_resident(x8)
        model._classification_threshold = float(torch.quantile(c8.float().sigmoid().reshape(x.size(0), -1)[0], 1.0 - 256.0 / c8.shape[1]))
        ms = _event_median_ms(lambda: model.detect_resident(x8, meta), 50)
        cn = model.detect_resident(x8, meta).counts.cpu().numpy()
        out['frames_uint8_1080p'] = dict(workload='WIDERFACE_LFD_S 8 x 1920x1080 uint8 NHWC resident (normalised in the stem), ~256 candidates per '
                                                  'image, IoU 0.4, forward + decode + NMS (one HIP graph, one batch in flight)',
                                         ms_per_step=round(ms, 4), images_per_s=round(x.size(0) / ms * 1e3, 1),
                                         candidates_per_image=float(cn[:, 0].mean()), kept_per_image=float(cn[:, 1].mean()))
        # ... and as the reference's LFD.forward receives it: normalised fp32 NCHW (lfd.py:511-542)
        x32 = torch.rand((x.size(0), 3, x.size(1), x.size(2)), device=dev, generator=gen) * 2 - 1
        c32, _ = model.forward_resident(x32)
        model._classification_threshold = float(torch.quantile(c32.float().sigmoid().reshape(x.size(0), -1)[0], 1.0 - 256.0 / c32.shape[1]))
        ms = _event_median_ms(lambda: model.detect_resident(x32, meta), 50)
        out['frames_fp32_nchw_1080p'] = dict(workload='WIDERFACE_LFD_S 8 x 3 x 1080 x 1920 fp32 NCHW resident, ~256 candidates per image, IoU 0.4, '
                                                      'forward + decode + NMS (one HIP graph, one batch in flight)',
                                             ms_per_step=round(ms, 4), images_per_s=round(x.size(0) / ms * 1e3, 1))
        del x8, x32, c8, c32
    finally:
        model._classification_threshold, model._nms_cfg, model.use_graph = keep[0], dict(keep[1]), keep[2]
    return out


def precise_breakdown(model, x, dev, reps=20, timer=None):
    """Per-launch HIP-event times of the planes plan (eager, one launch at a time on the current stream), grouped by kernel
    class, with the ALGORITHMIC conv FLOPs (2 MAC; the hi/lo products are an implementation detail: x3 issued) and bytes
    (both planes of every tensor read / written once) -> roofline entries of the tolerance-compliant mode."""
    from lfd_amd import engine, engine_p2, engine_p32
    plan = engine_p32.get_plan(model, x.device)
    if not isinstance(plan, engine_p2.PlanesPlan):
        return None
    fmt, n, h, w = engine._input_format(x)
    st = plan.state_for(n, h, w, 0)
    plan.run(x, fmt, st)
    torch.cuda.synchronize()
    groups = {}
    fused_stem = (plan.stem2x is not None and engine_p2._stem2x_enabled() and fmt == 1 and w % 8 == 0 and x.data_ptr() % 16 == 0)
    by_levels = plan.level_groups is not None and engine_p2._levels_enabled()

    def cost(o):
        if o.kind == 'stem':
            oh, ow = st.dims[o.dst]
            c = o.channels
            return (2.0 * n * oh * ow * (27 * c + c * c), x.numel() * x.element_size() + 4.0 * n * oh * ow * c,
                    'stem pair 1: conv3x3 s2 (3->%d) + conv1x1 (k_pl_stem)' % c)
        ih, iw = st.dims[o.src]           # (not st.bufs[...]: plane buffers are allocated at their first launch)
        oh, ow = (ih + o.stride - 1) // o.stride, (iw + o.stride - 1) // o.stride
        fl = 2.0 * n * oh * ow * o.cin * o.cout * o.ks * o.ks
        by = 4.0 * n * ih * iw * o.cin
        if o.tail is not None:
            fl += 2.0 * n * oh * ow * o.cout * o.cout
        if o.ds is not None:
            fl += 2.0 * n * oh * ow * o.cin * o.cout
            by += 4.0 * n * oh * ow * o.cout
        if o.res is not None:
            by += 4.0 * n * oh * ow * o.cout
        by += 4.0 * n * oh * ow * (o.cout if o.out_mode != 2 else o.f_c0 + o.f_c1)
        if o.ks == 3 and o.stride == 1 and o.cin == 64:
            key = 'conv3x3 s1 64->64 (+ residual) (k_pl_c3p)'
        elif o.ks == 3 and o.stride == 2 and o.tail is not None:
            key = 'stem pair 2: conv3x3 s2 + conv1x1 (k_pl_conv<64,3,2,TAIL>)'
        elif o.ks == 3 and o.stride == 2:
            key = 'stage entry: conv3x3 s2 + 1x1 s2 identity branch (k_pl_conv<.,3,2,DS>)'
        elif o.ks == 3:
            key = 'conv3x3 s1 128->128 (k_pl_conv<128,3,1>)'
        else:
            key = 'neck + head 1x1 convs, GroupNorm in the consumer (k_pl_conv<.,1,1>)'
        return fl, by, key

    # launch units: (key, flops, bytes, launches, callable)
    units, i = [], 0
    while i < len(plan.ops):
        if fused_stem and i == 0:
            (f0, b0, _), (f1, b1, _) = cost(plan.ops[0]), cost(plan.ops[1])
            mid = 4.0 * n * st.dims[plan.ops[0].dst][0] * st.dims[plan.ops[0].dst][1] * plan.ops[0].channels
            units.append(('whole stem: conv3x3 s2 (3->64) + 1x1 + conv3x3 s2 + 1x1, pair-1 output never in HBM (k_pl_stem2xs: row stream, producer + consumer waves)', f0 + f1,
                          b0 + b1 - 2 * mid, 1, lambda: plan._launch(x, fmt, st, [0, 1])))
            i = 2
            continue
        if by_levels and i == plan.head_start:
            fl = by = 0.0
            for o in plan.ops[i:]:
                f, b, _ = cost(o)
                fl, by = fl + f, by + b
            flat = engine_p2._flat_head_enabled() and plan._flat_head_modes() is not None
            units.append(('neck + head 1x1 convs of all pyramid levels, GroupNorm in the consumer (%s)' % ('k_pl_head / k_pl_head_out: flat tiles, fp32 intermediates'
                                                                                                         if flat else 'k_pl_conv_ml'), fl, by, len(plan.level_groups),
                          lambda: plan._launch_levels(st)))
            break
        fl, by, key = cost(plan.ops[i])
        units.append((key, fl, by, 1, (lambda i=i: plan._launch(x, fmt, st, [i]))))
        i += 1
    for key, fl, by, nl, fn in units:
        if timer is not None:       # (tools/power_trace_r6.py: seconds of back-to-back launches under rocm-smi)
            us = timer(fn, key)
        else:
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in evs:
                e0.record()
                fn()
                e1.record()
            torch.cuda.synchronize()
            us = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs])) * 1e3
        g = groups.setdefault(key, dict(launches=0, time_us=0.0, flops=0.0, bytes=0.0))
        g['launches'] += nl
        g['time_us'] += us
        g['flops'] += fl
        g['bytes'] += by
    rows = []
    for key, g in sorted(groups.items(), key=lambda kv: -kv[1]['time_us']):
        tf, gbs = g['flops'] / g['time_us'] / 1e6, g['bytes'] / g['time_us'] / 1e3
        bound = 'mfma' if 3 * tf / MFMA_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS else 'hbm'      # (what the matrix pipe ISSUES decides the bound)
        rows.append(dict(kernel=key, launches=g['launches'], time_us_per_forward=round(g['time_us'], 1),
                         avg_launch_us=round(g['time_us'] / g['launches'], 2), bound=bound,
                         achieved=round(tf if bound == 'mfma' else gbs, 1), peak=MFMA_PEAK_TFLOPS if bound == 'mfma' else HBM_PEAK_GBS,
                         unit='TFLOP/s' if bound == 'mfma' else 'GB/s',
                         frac=round(tf / MFMA_PEAK_TFLOPS if bound == 'mfma' else gbs / HBM_PEAK_GBS, 3),
                         tflops_algorithmic=round(tf, 1), tflops_issued=round(3 * tf, 1), frac_mfma_issued=round(3 * tf / MFMA_PEAK_TFLOPS, 3),
                         gbs=round(gbs, 0), frac_hbm=round(gbs / HBM_PEAK_GBS, 3)))
    return rows


def compact_line(r):
    """the contract keys + roofline + cpu_baseline + serial / sustained figures + one-line summaries of the extra keys, < 2 KB"""
    keys = ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'precision_mode', 'ms_per_step_serial', 'images_per_s_serial', 'pipeline_depth', 'ranks_seen')
    c = {'metric': "images/sec WIDERFACE-S 1920x1080 bs=8 end-to-end inference, LFD.precision='fp32_storage' (raw logits within 1e-4 of fp32: "
                   "inside north_star's 1e-3; the 'fp16' mode, sigma 2.5e-3, under fp16_mode)"}
    c.update({k: r[k] for k in keys if k in r})
    su = r.get('images_per_s_sustained') or {}
    c['images_per_s_sustained'] = {k: (su.get(k) or {}).get('images_per_s') for k in ('pipelined', 'serial')}
    cfg = r.get('config', {})
    c['config'] = {'workload': 'WIDERFACE_LFD_S inference bs=8/GPU 1920x1080 fp16 frames', 'global_batch': cfg.get('global_batch'),
                   'parallelism': cfg.get('parallelism'), 'hip_graph': cfg.get('hip_graph')}
    ev = r.get('step_ms_hip_events') or {}
    c['step_ms_hip_events'] = {k: ev.get(k) for k in ('median', 'p95')}
    rf = r.get('roofline') or {}
    c['roofline'] = {k: (rf.get(k)[:60] if k == 'kernel' and rf.get(k) else rf.get(k))
                     for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_mfma_issued', 'traffic', 'avg_launch_us')}
    for k in ('roofline_conv3x3_s1_64', 'roofline_backbone_3x3'):
        if k in r:
            c[k] = {'frac': r[k].get('frac'), 'frac_mfma_issued': r[k].get('frac_mfma_issued')}
    cb = r.get('cpu_baseline')
    c['cpu_baseline'] = None if not cb else {k: (cb.get(k)[:90] if k == 'sample' and cb.get(k) else cb.get(k))
                                               for k in ('value', 'unit', 'cores', 'kind', 'sample')}
    f = r.get('fp16_mode') or {}
    if f:
        c['fp16_mode'] = {'images_per_s': f.get('images_per_s'), 'ms_per_step': f.get('ms_per_step'), 'images_per_s_serial': f.get('images_per_s_serial'),
                          'sustained': ((f.get('images_per_s_sustained') or {}).get('pipelined') or {}).get('images_per_s'),
                          'frac_conv3x3': (f.get('roofline_conv3x3_s1_64') or {}).get('frac'),
                          'latency_bs1': {k: (v.get('p50') if isinstance(v, dict) else None) for k, v in (f.get('latency_bs1') or {}).items()
                                          if k in ('forward_ms', 'end_to_end_ms')}}
    cf = r.get('configs') or {}
    c['configs'] = {k: (cf.get(k) or {}).get('ms_per_step') for k in ('config3', 'config4') if k in cf}
    if 'frames_640x480' in cf:
        c['configs']['640x480_bs8'] = ((cf['frames_640x480'] or {}).get('bs8') or {}).get('ms_per_step')
    if 'frames_uint8_1080p' in cf:
        c['configs']['uint8_1080p_bs8_serial'] = (cf['frames_uint8_1080p'] or {}).get('ms_per_step')
    if 'frames_fp32_nchw_1080p' in cf:
        c['configs']['fp32_nchw_1080p_bs8_serial'] = (cf['frames_fp32_nchw_1080p'] or {}).get('ms_per_step')
    t = r.get('train') or {}
    c['train'] = {k: t.get(k) for k in ('ms_per_iter', 'n_gpus', 'ranks_seen', 'images_per_s', 'error') if k in t}
    lb = r.get('latency_bs1') or {}
    if lb:
        c['latency_bs1'] = {k: (v.get('p50') if isinstance(v, dict) else v) for k, v in lb.items() if k in ('forward_ms', 'end_to_end_ms')}
    c['full_line'] = 'the line above / gpurun_out/bench_full.json'
    return c


def latency_bs1(model, dev, iters=200):
    """p50 latency of ONE 1920x1080 frame (the second half of BASELINE.json's metric), frame resident in HBM, host-side
    wall clock around launch + synchronize per iteration like the reference's timing loop
    (lfd/deployment/tensorrt/inference_latency_evaluation.py:54-66; that one also copies the frame in and the outputs out):
    network forward only (what the reference's published latencies cover) and the end-to-end step (+ decode + NMS)."""
    gen = torch.Generator(device=dev).manual_seed(123)
    x1 = (torch.rand(1, H, W, 3, device=dev, generator=gen) * 2 - 1).half()
    meta1 = torch.tensor([[float(W), float(H), 1.0]], dtype=torch.float32, device=dev)
    out = {}
    # network only: the forward of this frame buffer captured once as a HIP graph (what detect_resident does for the whole
    # step; LFD.forward_resident itself re-validates the weight plan on every call, ~0.15 ms of Python at bs 1)
    keep = model.use_graph
    model.use_graph = False
    model.forward_resident(x1)
    torch.cuda.synchronize()
    fwd_graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(fwd_graph):
        model.forward_resident(x1)
    model.use_graph = keep
    for name, fn in (('forward_ms', fwd_graph.replay), ('end_to_end_ms', lambda: model.detect_resident(x1, meta1))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts)) * 1e3
        out[name] = {'p50': round(float(ts[len(ts) // 2]), 4), 'p90': round(float(ts[int(len(ts) * 0.9)]), 4),
                     'min': round(float(ts[0]), 4)}
    out['iterations'] = iters
    out['note'] = ('bs 1, frame resident in HBM, one HIP graph per call; reference (other hardware, network only, incl. '
                   'H2D + D2H): 4.88 ms WIDERFACE-S 1080p TensorRT FP16 on RTX 2080Ti (BASELINE.md)')
    return out


def timed_region(model, xs, meta, args, P, streams, dev, sync_ranks):
    """The contract's timed region for the model's CURRENT precision mode: W untimed warm-up steps, then exactly K steps
    between barrier + synchronize pairs, every step one complete batch (forward + decode + NMS of 8 frames as one HIP graph).
    `--pipeline P` keeps P batches in flight: step i is enqueued on HIP stream i % P with buffer slot i % P (own activations,
    outputs and workspace; weights shared), so that the phases of one batch that leave most CUs idle (small-map stages,
    post-processing, launch gaps) overlap with the next batch's stem / blocks; P = 1 is the strictly serial replay, always
    reported beside it.  Step i works on frame buffer i % NBUF (distinct frames, 398 MB in rotation: cold reads) with buffer
    slot (i % NBUF) % P -- a frame buffer always meets the same slot, so there is one captured graph per frame buffer.
    Also measured, outside the timed region: the per-step HIP-event distribution of the serial replay, and SUSTAINED rates --
    >= 1 s of replays enqueued without any host synchronisation in between (round 6: the 20-step region is ~10-30 ms; lessons
    10 / 37: clocks settle lower once the chip runs gap-free for long)."""
    def step(i=0, serial=False):
        b = i % NBUF
        sl = b % P
        with torch.cuda.stream(streams[0 if serial else sl]):
            return model.detect_resident(xs[b], meta, slot=sl)      # one HIP graph per step unless --no-graph

    torch.cuda.synchronize()
    dets = [None] * NBUF
    for b in range(NBUF):               # graph capture per frame buffer (setup, not a warm-up step)
        dets[b] = step(b)
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.clock_warmup_s:      # setup: bring the clocks up (not counted as warm-up steps)
        for i in range(NBUF):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        dets[i % NBUF] = step(i)
    torch.cuda.synchronize()
    if sync_ranks and dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        dets[i % NBUF] = step(i)
    torch.cuda.synchronize()
    if sync_ranks and dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if sync_ranks and dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    for i in range(min(NBUF, args.steps + args.warmup), NBUF):
        dets[i] = step(i)
    torch.cuda.synchronize()
    snap = [(d_.counts.clone(), d_.dets.clone()) for d_ in dets]      # what the overlapped steps produced, per frame buffer
    # the strictly serial replay (one batch in flight), same number of steps: reported next to the headline
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, serial=True)
    torch.cuda.synchronize()
    dt_serial = time.perf_counter() - t0
    for b in range(NBUF):     # every frame buffer alone on one stream == what it produced with two batches in flight
        o = step(b, serial=True)
        torch.cuda.synchronize()
        assert torch.equal(o.counts, snap[b][0]), 'overlapped and serial steps disagree (buffer %d)' % b
        for j in range(BATCH):
            k_ = int(o.counts[j, 1])
            assert torch.equal(o.dets[j, :k_], snap[b][1][j, :k_]), 'overlapped and serial steps disagree (buffer %d)' % b
    counts = dets[0].counts.cpu().numpy()
    assert int(counts[:, 2].max()) == 0, 'candidate capacity overflow: raise --max-candidates'
    # per-step distribution with HIP events on the launch stream (SURVEY 8d protocol: >= 100 iterations, median + p95)
    nev = max(100, args.steps)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nev)]
    with torch.cuda.stream(streams[0]):
        for e0, e1 in evs:
            e0.record()
            model.detect_resident(xs[0], meta, slot=0)
            e1.record()
    torch.cuda.synchronize()
    ev_ms = np.sort(np.array([e0.elapsed_time(e1) for e0, e1 in evs]))
    # sustained: >= args.sustained_s seconds of device time enqueued back to back, NO host synchronisation inside
    sustained = None
    if args.sustained_s > 0 and model.use_graph:
        sustained = {}
        for name, serial, per_step in (('pipelined', False, dt / args.steps), ('serial', True, dt_serial / args.steps)):
            n = max(args.steps, int(args.sustained_s / per_step) + 1)
            torch.cuda.synchronize()
            w0 = time.time()
            t0 = time.perf_counter()
            for i in range(n):
                step(i, serial=serial)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            sustained[name] = dict(images_per_s=round(BATCH * n / el, 1), ms_per_step=round(el / n * 1e3, 4), steps=n, seconds=round(el, 2),
                                   wall_clock_window=[round(w0, 3), round(w0 + el, 3)])      # (tools/power_trace_r6.py joins rocm-smi samples on it)
        sustained['note'] = 'gap-free replays for >= %.1f s of device time (no host synchronisation inside), this rank' % args.sustained_s
    return dict(dt=dt, dt_serial=dt_serial, ev_ms=ev_ms, counts=counts, sustained=sustained)


def fp16_mode_report(model, fast, x, dev, args, P, world):
    """The 'fp16' precision mode on the headline workload (rounds 1-5 quoted `value` in it): sigma within 2.5e-3 of the fp32
    reference -- OUTSIDE north_star's 1e-3, hence not the headline -- measured by the same timed_region(); its kernel classes
    with live HIP-event times against the fp16 MFMA / HBM peaks."""
    from lfd_amd import engine
    dt, dt_serial, ev_ms = fast['dt'], fast['dt_serial'], fast['ev_ms']
    out = {'mode': "LFD.precision = 'fp16': fp16 MFMA operands, fp16 inter-layer storage, fp32 accumulation (fused stem / block / head kernels)",
           'parity': 'sigma(cls), sigma(reg) <= 2.5e-3 vs fp32 (measured <= 2.3e-3): outside the 1e-3 of north_star; NMS bit-exact on identical logits',
           'images_per_s': round(world * BATCH * args.steps / dt, 1), 'ms_per_step': round(dt / args.steps * 1e3, 4), 'pipeline_depth': P,
           'images_per_s_serial': round(world * BATCH * args.steps / dt_serial, 1), 'ms_per_step_serial': round(dt_serial / args.steps * 1e3, 4),
           'images_per_s_sustained': fast['sustained'],
           'step_ms_hip_events': {'median': round(float(ev_ms[len(ev_ms) // 2]), 4), 'p95': round(float(ev_ms[int(len(ev_ms) * 0.95)]), 4)},
           'mfma_tflops': round(world * 348.8 * args.steps / dt / 1e3, 1)}
    fmt, n, h, w = engine._input_format(x)
    plan = engine.get_plan(model, model._backbone, model._neck, model._head, dev)
    st = plan.state_for(n, h, w)
    br = kernel_breakdown(model, plan, st, x, fmt)
    tot = sum(c['time_us'] for c in br.values())
    rows = []
    for name, c in sorted(br.items(), key=lambda kv: -kv[1]['time_us']):
        tf = c['flops'] / c['time_us'] / 1e6
        gb = c['bytes'] / c['time_us'] / 1e3
        rows.append({'kernel': name, 'launches': c['launches'], 'time_us_per_forward': round(c['time_us'], 1),
                     'share': round(c['time_us'] / tot, 3), 'tflops': round(tf, 1), 'gbs': round(gb, 0),
                     'frac_mfma': round(tf / MFMA_PEAK_TFLOPS, 3), 'frac_hbm': round(gb / HBM_PEAK_GBS, 3)})
    dom = rows[0]
    # every 3x3 stride-1 64->64 conv of the backbone: the fused residual blocks + the stand-alone launches
    # (the 3x3 s1 conv inside the fused downsample block is not separable from its HBM-bound stride-2 conv: listed on its own)
    k33c = [c for nm, c in br.items() if nm.startswith('conv3x3_s1_64to64') or nm.startswith('fasterblock_fused')]
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    except Exception:
        pass
    if k33c:
        t33 = sum(c['time_us'] for c in k33c)
        f33 = sum(c['flops'] for c in k33c)
        nm33 = 'all conv3x3 s1 64->64 (fused residual blocks k_block64_rows / k_block64 + stand-alone k_conv)'
        out['roofline_conv3x3_s1_64'] = {'bound': 'mfma', 'achieved': round(f33 / t33 / 1e6, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                         'frac': round(f33 / t33 / 1e6 / MFMA_PEAK_TFLOPS, 3),
                                         'avg_launch_us': round(t33 / sum(c['launches'] for c in k33c), 2), 'traffic': pmc.get(nm33),
                                         'traffic_source': PMC_SOURCE,
                                         # information beside the contract's peak: what a pure MFMA loop holds on this part when
                                         # its operands change (power cap; profiles/r03_mfma_operand_power.txt)
                                         'pure_mfma_loop_random_operands_tflops': 1574.0, 'frac_of_that': round(f33 / t33 / 1e6 / 1574.0, 3),
                                         'power': 'at the socket power limit: 1375-1390 W, 2.0 GHz (profiles/r06_power_trace.json)'}
    # every launch of the backbone (stem, residual blocks, downsample blocks, 128-channel convs): north_star's "the 3x3
    # backbone convs" -- 34.3 of the backbone's 41.0 GFLOP per image are 3x3 taps, the 1x1s are fused into the same launches
    bbc = [c for nm, c in br.items() if not nm.startswith('neck+head')]
    tb, fb = sum(c['time_us'] for c in bbc), sum(c['flops'] for c in bbc)
    out['roofline_backbone_3x3'] = {'kernel': 'all backbone launches (fused stem + residual blocks + downsample blocks + 128-channel convs)',
                                    'bound': 'mfma', 'achieved': round(fb / tb / 1e6, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': round(fb / tb / 1e6 / MFMA_PEAK_TFLOPS, 3), 'time_us_per_forward': round(tb, 1),
                                    'launches': sum(c['launches'] for c in bbc), 'gflop_per_forward': round(fb / 1e9, 1)}
    if dom['frac_mfma'] >= dom['frac_hbm']:
        out['roofline'] = {'kernel': dom['kernel'], 'bound': 'mfma', 'achieved': round(dom['tflops'], 1), 'peak': MFMA_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': dom['frac_mfma'], 'traffic': pmc.get(dom['kernel']), 'traffic_source': PMC_SOURCE}
    else:
        out['roofline'] = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': dom['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': dom['frac_hbm'], 'traffic': pmc.get(dom['kernel']), 'traffic_source': PMC_SOURCE}
    out['roofline']['avg_launch_us'] = round(dom['time_us_per_forward'] / dom['launches'], 2)
    out['kernels'] = rows
    out['forward_sum_us'] = round(tot, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-latency', action='store_true')
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--no-siblings', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the extra key with BASELINE configs 3 and 4')
    ap.add_argument('--no-fp16', action='store_true', help="skip the extra key `fp16_mode` (the faster mode outside the 1e-3 tolerance)")
    ap.add_argument('--headline-mode', default='fp32_storage', choices=['fp32_storage', 'fp16'],
                    help="precision mode of the timed region; 'fp16' is for profiling that mode's kernels under rocprofv3 "
                         "(tools/collect_profiles.sh) -- the line then says so in `metric` and must not be read as the benchmark")
    ap.add_argument('--sustained-s', type=float, default=1.0, help='seconds of gap-free replays for `images_per_s_sustained` (0: skip)')
    ap.add_argument('--max-candidates', type=int, default=8192)
    ap.add_argument('--clock-warmup-s', type=float, default=0.3, help='untimed replays before the W warm-up steps: an idle MI355X needs '
                    'milliseconds to ramp its clocks (DESIGN 3, lesson 11)')
    ap.add_argument('--pipeline', type=int, default=2, help='batches in flight per GPU (HIP streams with their own buffers)')
    args = ap.parse_args()
    global HEADLINE_MODE
    HEADLINE_MODE = args.headline_mode
    if HEADLINE_MODE == 'fp16':
        args.no_fp16 = True

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher in the environment: become one (one process per GPU, RCCL rendezvous on 127.0.0.1) instead of silently
        # measuring a single rank
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ranks_seen = 1
    if world > 1 or os.environ.get('LFD_BENCH_FORCE_DIST') == '1':      # (FORCE: the N > 1 code path through RCCL on one GPU)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl')        # RCCL on ROCm
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)                   # every rank that really takes part adds itself
        ranks_seen = int(one.item())

    from lfd_amd import configs, engine
    model = configs.build_model(MODEL)
    configs.perturb_weights(model)
    model.eval().to(dev)
    model.use_graph = not args.no_graph
    # candidate capacity per image (an image with more candidates raises its overflow flag in `counts`, checked below)
    model.max_candidates = args.max_candidates
    gen = torch.Generator(device=dev).manual_seed(rank)
    xs = (torch.rand(NBUF, BATCH, H, W, 3, device=dev, generator=gen) * 2 - 1).half()   # NBUF resident batches of NHWC fp16 frames
    x = xs[0]
    meta = torch.tensor([[float(W), float(H), 1.0]] * BATCH, dtype=torch.float32, device=dev)

    with torch.no_grad():
        cls, reg = model.forward_resident(x)
        # score threshold giving ~TARGET_K candidates per image on these synthetic weights
        thr = float(torch.quantile(cls.float().sigmoid().reshape(BATCH, -1)[0], 1.0 - TARGET_K / cls.shape[1]))
        model._classification_threshold = thr
        model._nms_cfg = dict(type='nms', iou_thr=0.4)
        P = max(1, args.pipeline) if model.use_graph else 1
        streams = [torch.cuda.Stream(device=dev) for _ in range(P)]

        # ---- the contract's timed region, in the precision mode the metric is quoted in: 'fp32_storage', the mode inside
        #      north_star's tolerance (cls / bbox within 1e-3: raw logits <= 1e-4 here).  Round 6: rounds 1-5 quoted `value` in
        #      the 'fp16' mode (sigma within 2.5e-3) -- that figure is measured the same way below and reported as `fp16_mode`.
        model.precision = HEADLINE_MODE
        hl = timed_region(model, xs, meta, args, P, streams, dev, sync_ranks=True)
        model.precision = 'fp16'
        fast = None
        if world == 1 and not args.no_fp16:
            fast = timed_region(model, xs, meta, args, P, streams, dev, sync_ranks=False)

        result = None
        if rank == 0:
            dt, counts = hl['dt'], hl['counts']
            value = world * BATCH * args.steps / dt
            ev_ms = hl['ev_ms']
            result = {
                'metric': "images/sec WIDERFACE-S 1920x1080 bs=8 end-to-end inference (forward + decode + NMS), LFD.precision='fp32_storage': "
                          "the mode inside north_star's tolerance (cls / bbox raw logits within 1e-4 of the fp32 reference, sigma within "
                          "1e-3; kept indices bit-exact on identical logits); the faster 'fp16' mode (sigma within 2.5e-3, OUTSIDE the 1e-3) "
                          "is reported under `fp16_mode`, never as `value`",
                'parity_gates': {"fp32_storage (this value)": 'raw logits <= 1e-4 (measured <= 8.5e-6), sigma <= 1e-3 vs the fp32 reference at '
                                                              "configs 2 / 3 / 4 (tests/test_gpu_precise.py); the reference's end-to-end rows "
                                                              'reproduced with 0 unmatched, kept (point, class) lists identical (tests/test_gpu_end2end.py)',
                                 "fp16 (`fp16_mode`)": 'sigma(cls), sigma(reg) <= 2.5e-3 vs fp32 (measured <= 2.3e-3), raw logits <= 2e-2; '
                                                       'kept indices bit-exact on identical logits (tests/test_gpu_parity_fullsize.py)'},
                'value': round(value, 1), 'unit': 'images/s', 'n_gpus': world, 'ranks_seen': ranks_seen, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'precision_mode': HEADLINE_MODE,
                'clock_warmup_s': args.clock_warmup_s, 'pipeline_depth': P, 'ms_per_step_serial': round(hl['dt_serial'] / args.steps * 1e3, 4),
                'images_per_s_serial': round(world * BATCH * args.steps / hl['dt_serial'], 1),
                'images_per_s_sustained': hl['sustained'],
                'step_ms_hip_events': {'median': round(float(ev_ms[len(ev_ms) // 2]), 4), 'p95': round(float(ev_ms[int(len(ev_ms) * 0.95)]), 4),
                                       'min': round(float(ev_ms[0]), 4), 'iterations': int(len(ev_ms)),
                                       'note': 'latency of ONE step replayed alone (serial), HIP events on its stream'},
                'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
                'dtype_note': 'fp16 MFMA operands as hi + 2^-11 lo planes (three products per k-step), fp32 accumulation and epilogues',
                'mfma_tflops_issued': round(world * 3 * 348.8 * args.steps / dt / 1e3, 1),
                'config': {'workload': 'WIDERFACE_LFD_S inference bs=8/GPU 1920x1080 fp16 NHWC frames resident in HBM: '
                                       'backbone+neck+head (HIP MFMA convs on hi/lo fp16 planes) + decode + threshold + NMS, results on device',
                           'global_batch': world * BATCH, 'points_per_image': int(cls.shape[1]),
                           'input_buffers': '%d distinct resident batches rotated (%.0f MB > 256 MiB Infinity Cache)' % (NBUF, NBUF * x.numel() * 2 / 1e6),
                           'candidates_per_image': float(counts[:, 0].mean()), 'kept_per_image': float(counts[:, 1].mean()),
                           'score_thr': thr, 'iou_thr': 0.4, 'max_candidates': int(model.max_candidates), 'parallelism': 'image-parallel x%d, no collective; %d batches in flight per GPU (HIP streams)' % (world, P),
                           'hip_graph': bool(model.use_graph),
                           'weights': 'random init (seed 666) + synthetic BN/GN/Scale perturbation (no checkpoints offline)'},
            }
            if HEADLINE_MODE == 'fp16':
                result['metric'] = ("PROFILING RUN (--headline-mode fp16), not the benchmark: images/sec WIDERFACE-S 1920x1080 bs=8 in "
                                    "LFD.precision='fp16' (sigma within 2.5e-3: outside north_star's 1e-3)")
                result['mfma_tflops_issued'] = round(world * 348.8 * args.steps / dt / 1e3, 1)
                result['fp16_mode'] = fp16_mode_report(model, hl, x, dev, args, P, world)
                for k_ in ('roofline', 'roofline_conv3x3_s1_64', 'roofline_backbone_3x3', 'kernels', 'forward_sum_us'):
                    if k_ in result['fp16_mode']:
                        result[k_] = result['fp16_mode'][k_]
            # ---- roofline of the headline mode's kernel classes (live HIP-event timing on the launch stream, eager launches)
            try:
                if HEADLINE_MODE == 'fp16':
                    raise StopIteration
                model.precision = HEADLINE_MODE
                prow = precise_breakdown(model, x, dev)
                model.precision = 'fp16'
                if prow:
                    ptraffic = {}
                    try:
                        ptraffic = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic_precise.json')))
                    except Exception:
                        pass
                    dom = prow[0]
                    result['roofline'] = {'kernel': dom['kernel'], 'bound': dom['bound'], 'achieved': dom['achieved'], 'peak': dom['peak'],
                                          'unit': dom['unit'], 'frac': dom['frac'], 'avg_launch_us': dom['avg_launch_us'],
                                          'frac_mfma_issued': dom['frac_mfma_issued'], 'traffic': ptraffic.get(dom['kernel']),
                                          'traffic_source': PMC_SOURCE_PRECISE,
                                          'note': 'ALGORITHMIC flops (2 MAC per tap) / bytes over the live HIP-event duration; the three hi/lo '
                                                  'MFMA products per k-step ISSUE 3 x the algorithmic flops (frac_mfma_issued)'}
                    for r_ in prow:
                        r_['traffic'] = ptraffic.get(r_['kernel'])
                    c3 = [r_ for r_ in prow if r_['kernel'].startswith('conv3x3 s1 64->64')]
                    if c3:
                        result['roofline_conv3x3_s1_64'] = {'bound': 'mfma', 'achieved': c3[0]['tflops_algorithmic'], 'peak': MFMA_PEAK_TFLOPS,
                                                            'unit': 'TFLOP/s', 'frac': round(c3[0]['tflops_algorithmic'] / MFMA_PEAK_TFLOPS, 3),
                                                            'frac_mfma_issued': c3[0]['frac_mfma_issued'], 'avg_launch_us': c3[0]['avg_launch_us'],
                                                            'launches': c3[0]['launches'], 'traffic': c3[0]['traffic'],
                                                            'power': 'at the socket power limit: 1350-1400 W, 2.0 GHz (profiles/r06_power_trace.json)'}
                    bb = [r_ for r_ in prow if not r_['kernel'].startswith('neck + head')]
                    tb = sum(r_['time_us_per_forward'] for r_ in bb)
                    fb = sum(r_['tflops_algorithmic'] * r_['time_us_per_forward'] for r_ in bb)      # TFLOP/s x us = MFLOP
                    result['roofline_backbone_3x3'] = {'kernel': 'all backbone launches of the headline mode (k_pl_stem2xs, k_pl_c3p, stage entries, 128-channel convs)',
                                                       'bound': 'mfma', 'achieved': round(fb / tb, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                                       'frac': round(fb / tb / MFMA_PEAK_TFLOPS, 3), 'frac_mfma_issued': round(3 * fb / tb / MFMA_PEAK_TFLOPS, 3),
                                                       'time_us_per_forward': round(tb, 1), 'launches': sum(r_['launches'] for r_ in bb)}
                    result['kernels'] = prow
                    result['forward_sum_us'] = round(sum(r_['time_us_per_forward'] for r_ in prow), 1)
            except StopIteration:
                pass
            except Exception as e:
                model.precision = 'fp16'
                result['roofline'] = {'error': repr(e)}
            if fast is not None:
                result['fp16_mode'] = fp16_mode_report(model, fast, x, dev, args, P, world)
    if rank == 0 and world == 1 and not args.no_latency:
        with torch.no_grad():
            model.precision = HEADLINE_MODE
            result['latency_bs1'] = latency_bs1(model, dev)
            model.precision = 'fp16'
            if 'fp16_mode' in result:
                result['fp16_mode']['latency_bs1'] = latency_bs1(model, dev)
    if rank == 0 and world == 1 and not args.no_configs:
        try:
            with torch.no_grad():
                result['configs'] = other_configs_bench(dev)
                model.precision = HEADLINE_MODE
                cls_p, _ = model.forward_resident(x)
                result['configs'].update(stress_and_small_frame_bench(model, x, meta, dev, cls_p))
                model.precision = 'fp16'
                if 'fp16_mode' in result:
                    result['fp16_mode']['configs'] = stress_and_small_frame_bench(model, x, meta, dev, cls)
        except Exception as e:
            model.precision = 'fp16'
            result['configs'] = {'error': repr(e)}
    if rank == 0:
        # (rounds 3-5 reported the tolerance-compliant mode under this key; it IS the headline now -- alias kept for readers of the old layout)
        lb_ = (result.get('latency_bs1') or {}).get('end_to_end_ms') or {}
        result['precise'] = {'same_as': 'the headline keys of this line (value / ms_per_step / ms_per_step_serial / images_per_s_sustained / latency_bs1 / configs)',
                             'images_per_s_bs8': result['value'], 'ms_per_step_bs8': result['ms_per_step'], 'pipeline_depth': result['pipeline_depth'],
                             'images_per_s_bs8_serial': result['images_per_s_serial'], 'ms_per_step_bs8_serial': result['ms_per_step_serial'],
                             'end_to_end_bs1_ms': {'p50': lb_.get('p50'), 'min': lb_.get('min')}}
    if not args.no_train:
        # every rank runs the training leg (the DDP iteration has collectives); rank 0 reports
        try:
            tr = train_bench(dev, world=world, rank=rank)
        except Exception as e:       # the inference line must not be lost to the extra key
            tr = {'error': repr(e)}
        if rank == 0:
            result['train'] = tr
    if rank == 0 and world == 1 and not args.no_siblings:
        try:
            result['siblings'] = siblings_bench(dev)
        except Exception as e:       # an extra key, like `train`
            result['siblings'] = {'error': repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline()
        else:
            result['cpu_baseline'] = None
        try:      # RCCL writes its banner through C stdio: push it out BEFORE the contract lines (it must not end the output)
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)
        # the driver keeps a 2 KB tail of stdout: the contract line once more, compact, LAST -- headline keys, `roofline`,
        # `cpu_baseline`, the serial / HIP-event figures and one-line summaries of the extra keys (everything is in the long
        # line above and in gpurun_out/bench_full.json)
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            json.dump(result, open(os.path.join(ROOT, 'gpurun_out', 'bench_full.json'), 'w'))
        except Exception:
            pass
        print(json.dumps(compact_line(result)), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
