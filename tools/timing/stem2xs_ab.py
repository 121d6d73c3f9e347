"""k_pl_stem2xs (csrc/planes_stem2xs.hip: the fused plane stem as a row stream with producer and consumer waves) against
k_pl_stem2x (tiles, one wave per SIMD): results on small and odd shapes vs float64, and launch times at 8 x 1080p, interleaved
in one session.
    python tools/timing/stem2xs_ab.py [quick]          (LFD_HIP_LIB=... for a variant build, tools/ab_build.sh)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from lfd_amd import ops, _lib, engine_p2  # noqa: E402
from lfd_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

L = lib()
dev = torch.device('cuda')
z = ops.zero_line(dev)
c = 64


def weights(seed):
    g = torch.Generator().manual_seed(seed)
    w1, b1 = torch.randn(c, 3, 3, 3, generator=g) * 0.3, torch.randn(c, generator=g)
    w2, b2 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    w3, b3 = torch.randn(c, c, 3, 3, generator=g) * (1.0 / (9 * c) ** 0.5), torch.randn(c, generator=g)
    w4, b4 = torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5), torch.randn(c, generator=g)
    keep = [engine_p2.pack_planes_stem2x_weight(w1, b1).cuda(), engine_p2.pack_planes_stem2x_tail_weight(w2).cuda(),
            engine_p2._pad_bias(b2).cuda(), engine_p2.pack_planes_weight(w3).cuda(), engine_p2._pad_bias(b3, 128).cuda(),
            engine_p2.pack_planes_weight(w4).cuda(), engine_p2._pad_bias(b4, 128).cuda()]
    return (w1, b1, w2, b2, w3, b3, w4, b4), keep


def conv64(x, w, b, ks, stride):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=ks // 2).permute(0, 2, 3, 1)
    return y.relu()


def rt(t):
    return engine_p2.from_planes(engine_p2.to_planes(t.float())).double()


def run(xd, n, h, w, keep, out):
    check(L.lfd_pl_stem2x(ptr(xd), 1, n, h, w, *[ptr(k) for k in keep], ptr(out), out[0].numel(), ptr(z), stream_ptr()), 'lfd_pl_stem2x')


ok = True
for (n, h, w) in [(2, 75, 136), (1, 270, 480), (2, 37, 128), (1, 8, 8), (3, 129, 264), (1, 16, 2000), (2, 1080, 64)]:
    ws, keep = weights(h * 3 + w)
    g = torch.Generator().manual_seed(h)
    x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half()
    y = conv64(rt(x.float()), ws[0], ws[1], 3, 2)
    y = conv64(rt(y), ws[2], ws[3], 1, 1)
    y = conv64(rt(y), ws[4], ws[5], 3, 2)
    ref = conv64(rt(y), ws[6], ws[7], 1, 1)
    oh, ow = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
    xd = x.cuda()
    errs = {}
    for mode in (0, 1):
        _lib.tune('PL_STEM', mode)
        out = torch.full((2, n, oh, ow, c), float('nan'), dtype=torch.float16, device=dev)
        run(xd, n, h, w, keep, out); torch.cuda.synchronize()
        got = engine_p2.from_planes(out.cpu()).double()
        bad = torch.isnan(got).sum().item()
        errs[mode] = (float((got - ref).abs().nan_to_num(0).max()), bad)
    mag = float(ref.abs().max())
    good = errs[1][1] == 0 and errs[1][0] <= 4e-6 * max(1.0, mag)
    ok = ok and good
    print('%dx%dx%d: err tiles %.2e (%d nan)  stream %.2e (%d nan)  (max |y| %.2f)  %s' % (n, h, w, errs[0][0], errs[0][1], errs[1][0], errs[1][1], mag,
                                                                                  'ok' if good else 'FAIL'), flush=True)
print('CORRECT' if ok else 'WRONG', flush=True)

if 'sizes' in sys.argv:
    # launch time by batch size and frame size (events around 50 back-to-back launches)
    for (n, h, w) in [(1, 1080, 1920), (2, 1080, 1920), (4, 1080, 1920), (1, 480, 640), (8, 480, 640), (1, 2160, 3840)]:
        ws, keep = weights(1)
        xs = [(torch.rand(n, h, w, 3, device=dev) * 2 - 1).half() for _ in range(2)]
        oh, ow = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
        out = torch.empty((2, n, oh, ow, c), dtype=torch.float16, device=dev)
        res = {}
        for rep in range(2):
            for mode in (0, 1):
                _lib.tune('PL_STEM', mode)
                for i in range(5):
                    run(xs[i & 1], n, h, w, keep, out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(50):
                    run(xs[i & 1], n, h, w, keep, out)
                e1.record(); torch.cuda.synchronize()
                res[mode] = e0.elapsed_time(e1) * 20
        print('%d x %d x %d: tiles %.1f us  stream %.1f us' % (n, h, w, res[0], res[1]), flush=True)

if 'quick' not in sys.argv:
    n, h, w = 8, 1080, 1920
    ws, keep = weights(1)
    xs = [(torch.rand(n, h, w, 3, device=dev) * 2 - 1).half() for _ in range(4)]
    out = torch.empty((2, n, 270, 480, c), dtype=torch.float16, device=dev)
    outs = {}
    for rep in range(3):
        for mode in (0, 1):
            _lib.tune('PL_STEM', mode)
            for i in range(3):
                run(xs[i & 3], n, h, w, keep, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                run(xs[i & 3], n, h, w, keep, out)
            e1.record(); torch.cuda.synchronize()
            print('rep %d  %s: %.1f us' % (rep, 'stream' if mode else 'tiles ', e0.elapsed_time(e1) * 50), flush=True)
            run(xs[0], n, h, w, keep, out); torch.cuda.synchronize()
            outs[mode] = engine_p2.from_planes(out.cpu())
    print('max |tiles - stream| at 8 x 1080p: %.2e' % float((outs[0] - outs[1]).abs().max()))
    if hasattr(L, 'lfd_debug_pl_stem2xs_timing'):
        import numpy as np
        _lib.tune('PL_STEM', 1)
        for i in range(20):            # (the stamps of the last launch of a back-to-back series: the clock the series settles at)
            run(xs[i & 3], n, h, w, keep, out)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 136)()
        L.lfd_debug_pl_stem2xs_timing(buf)
        raw = np.array(list(buf), dtype=np.int64)
        t = raw[:128].reshape(8, 16)
        cyc, rtm = raw[130] - raw[128], raw[131] - raw[129]
        print('workgroup 0: %d slots, %d cycles (%.0f per slot) in %.1f us -> %.2f GHz' % (raw[132], cyc, cyc / max(raw[132], 1), rtm / 100.0, cyc / (rtm * 10.0)))
        print('stamps (slots 8..15 of workgroup 0; cycles relative to the producer\'s slot start):')
        print('   P: group 1 done, all done, barrier passed | kh0: start, -, contraction done, barrier passed | kh1: start, chained 1x1 staged, contraction done, barrier passed')
        for r in range(8):
            print('  ', ' '.join('%6d' % (t[r, i] - t[r, 0]) for i in range(1, 12)), '   slot length %d' % (t[r + 1, 0] - t[r, 0] if r < 7 else 0))
_lib.tune('PL_STEM', 1)

if 'power' in sys.argv:
    # socket power and clock of both kernels, back to back for 2 s each (rocm-smi polled by a child process)
    import time
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import power_trace_r6 as ptr6
    ptr6.start_poller()
    time.sleep(1.0)
    for mode in (0, 1, 0, 1):
        _lib.tune('PL_STEM', mode)
        run(xs[0], n, h, w, keep, out); torch.cuda.synchronize()
        t0 = time.time(); cnt = 0
        while time.time() - t0 < 2.0:
            for i in range(100):
                run(xs[i & 3], n, h, w, keep, out)
            torch.cuda.synchronize(); cnt += 100
        dt = time.time() - t0
        ptr6.windows['m'] = (t0, t0 + dt)
        time.sleep(0.1)
        print('%s: %.1f us per launch back to back for %.1f s  %s' % ('stream' if mode else 'tiles ', dt / cnt * 1e6, dt, ptr6.summarize('m')), flush=True)
        time.sleep(0.5)
    ptr6.stop_poller()
    _lib.tune('PL_STEM', 1)

if 'soak' in sys.argv:
    # the stream kernel is deterministic by construction (no atomics, fixed hand-over order): any LDS hazard shows as a bit that differs
    # between launches on the same input -- 8 x 1080p, 300 launches interleaved with other work on a second stream
    _lib.tune('PL_STEM', 1)
    n, h, w = 8, 1080, 1920
    ws, keep = weights(5)
    x = (torch.rand(n, h, w, 3, device=dev) * 2 - 1).half()
    out = torch.empty((2, n, 270, 480, c), dtype=torch.float16, device=dev)
    run(x, n, h, w, keep, out); torch.cuda.synchronize()
    ref = out.clone()
    side = torch.cuda.Stream()
    junk = torch.rand(4096, 4096, device=dev)
    bad = 0
    for i in range(300):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk = (junk @ junk).clamp_(-1, 1)
        out.fill_(float('nan'))
        run(x, n, h, w, keep, out)
        torch.cuda.synchronize()
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            bad += 1
    print('soak: %d of 300 launches differ from the first one bit for bit  %s' % (bad, 'DETERMINISTIC' if bad == 0 else 'RACE'), flush=True)
    for (n, h, w) in [(3, 129, 264), (1, 2160, 3840), (5, 66, 72)]:
        x = (torch.rand(n, h, w, 3, device=dev) * 2 - 1).half()
        oh, ow = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
        out = torch.empty((2, n, oh, ow, c), dtype=torch.float16, device=dev)
        run(x, n, h, w, keep, out); torch.cuda.synchronize()
        ref = out.clone()
        bad = 0
        for i in range(100):
            out.fill_(float('nan'))
            run(x, n, h, w, keep, out); torch.cuda.synchronize()
            bad += int(not torch.equal(out.view(torch.int16), ref.view(torch.int16)))
        print('soak %d x %d x %d: %d of 100 differ' % (n, h, w, bad), flush=True)

if 'f32' in sys.argv:
    # fp32 NCHW frames (what the reference's LFD.forward receives)
    n, h, w = 8, 1080, 1920
    ws, keep = weights(2)
    xs32 = [torch.rand(n, 3, h, w, device=dev) * 2 - 1 for _ in range(2)]
    out = torch.empty((2, n, 270, 480, c), dtype=torch.float16, device=dev)
    outs = {}

    def run32(x):
        check(L.lfd_pl_stem2x(ptr(x), 0, n, h, w, *[ptr(k) for k in keep], ptr(out), out[0].numel(), ptr(z), stream_ptr()), 'lfd_pl_stem2x')
    for rep in range(2):
        for mode in (0, 1):
            _lib.tune('PL_STEM', mode)
            for i in range(3):
                run32(xs32[i & 1])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                run32(xs32[i & 1])
            e1.record(); torch.cuda.synchronize()
            print('fp32 NCHW frames rep %d  %s: %.1f us' % (rep, 'stream' if mode else 'tiles ', e0.elapsed_time(e1) * 50), flush=True)
            run32(xs32[0]); torch.cuda.synchronize()
            outs[mode] = engine_p2.from_planes(out.cpu())
    print('fp32 NCHW frames: max |tiles - stream| at 8 x 1080p: %.2e' % float((outs[0] - outs[1]).abs().max()))
    _lib.tune('PL_STEM', 1)

if 'u8' in sys.argv:
    # uint8 NHWC frames (simple_normalize in the stem): the stream kernel's byte table against the tile kernel's load path
    n, h, w = 8, 1080, 1920
    ws, keep = weights(2)
    xs8 = [torch.randint(0, 256, (n, h, w, 3), device=dev, dtype=torch.uint8) for _ in range(2)]
    out = torch.empty((2, n, 270, 480, c), dtype=torch.float16, device=dev)
    outs = {}

    def run8(x):
        check(L.lfd_pl_stem2x(ptr(x), 2, n, h, w, *[ptr(k) for k in keep], ptr(out), out[0].numel(), ptr(z), stream_ptr()), 'lfd_pl_stem2x')
    for rep in range(2):
        for mode in (0, 1):
            _lib.tune('PL_STEM', mode)
            for i in range(3):
                run8(xs8[i & 1])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                run8(xs8[i & 1])
            e1.record(); torch.cuda.synchronize()
            print('uint8 frames rep %d  %s: %.1f us' % (rep, 'stream' if mode else 'tiles ', e0.elapsed_time(e1) * 50), flush=True)
            run8(xs8[0]); torch.cuda.synchronize()
            outs[mode] = engine_p2.from_planes(out.cpu())
    print('uint8 frames: max |tiles - stream| at 8 x 1080p: %.2e' % float((outs[0] - outs[1]).abs().max()))
    _lib.tune('PL_STEM', 1)
