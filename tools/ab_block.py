"""A/B timing of the fused FasterBlock launch (csrc/block.hip) against the two conv launches it replaces, HIP events on the
launch stream, at the three 64-channel map sizes of WIDERFACE_LFD_S @1080p batch 8.  Prints one JSON line per shape and
writes gpurun_out/ab_block.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch  # noqa: E402
from lfd_amd import ops  # noqa: E402


def _once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def timed_pair(fa, fb, reps=50, rounds=7):
    """A/B with a WARM GPU: the clocks of an idle MI355X take milliseconds to ramp (a 1 ms benchmark right after process
    start measured 10-40x too slow), so run ~0.3 s of the two candidates first, then interleave rounds and take medians."""
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(20):
            fa()
            fb()
        torch.cuda.synchronize()
    ta, tb = [], []
    for _ in range(rounds):
        ta.append(_once(fa, reps))
        tb.append(_once(fb, reps))
    ta.sort()
    tb.sort()
    return ta[len(ta) // 2], tb[len(tb) // 2]


def main():
    g = torch.Generator().manual_seed(0)
    w1 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
    w2 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
    b1, b2 = torch.randn(64, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
    out = []
    for (n, h, w) in ((8, 135, 240), (8, 68, 120), (8, 34, 60), (1, 135, 240), (32, 135, 240), (4, 180, 320), (1, 540, 960)):
        x = (torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda()
        mid, y2, y1 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

        def two():
            ops.conv2d_nhwc(x, w1, b1, 64, 64, 3, 1, True, out=mid)
            ops.conv2d_nhwc(mid, w2, b2, 64, 64, 3, 1, True, residual=x, out=y2)

        def one():
            ops.fasterblock_fused(x, w1, b1, w2, b2, out=y1)

        t2, t1 = timed_pair(two, one)
        gf = 2 * 2.0 * n * h * w * 64 * 64 * 9 / 1e9
        rec = dict(shape=[n, h, w], two_launch_us=round(t2, 2), fused_us=round(t1, 2), speedup=round(t2 / t1, 3),
                   fused_tflops=round(gf / t1 * 1e3, 1), two_tflops=round(gf / t2 * 1e3, 1),
                   fused_frac_mfma=round(gf / t1 * 1e3 / 2500.0, 3), identical=bool(torch.equal(y1, y2)))
        print(json.dumps(rec))
        out.append(rec)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'ab_block.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
