"""fused downsample block vs the two-launch path: per-shape kernel time from HIP graphs of 20 chained repeats.
Usage: python tools/timing/down_time.py [n h w ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')): sys.path.insert(0, p)
from lfd_amd import ops
shapes = [(8, 270, 480), (8, 135, 240), (8, 68, 120), (1, 270, 480), (1, 135, 240), (32, 160, 160)]
if len(sys.argv) > 3:
    a = list(map(int, sys.argv[1:])); shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
g = torch.Generator().manual_seed(0)
w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24); wd = (torch.randn(64, 64, 1, 1, generator=g) / 8); w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24)
p1, pd, p2 = (ops.pack_conv_weight(t).cuda() for t in (w1, wd, w2))
b = (torch.randn(64, generator=g) * 0.1).cuda()
REP = 20
def timed(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(REP): fn()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / REP)
    return min(ts)
for (n, h, w) in shapes:
    xs = [(torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda() for _ in range(4)]
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(n, oh, ow, 64, dtype=torch.float16, device='cuda')
    k = [0]
    def two():
        x = xs[k[0] % 4]; k[0] += 1
        y1, idn = ops.conv2d_downsample_nhwc(x, p1, b, pd, b)
        ops.conv2d_nhwc(y1, p2, b, 64, 64, 3, 1, True, residual=idn, out=out)
    def one():
        x = xs[k[0] % 4]; k[0] += 1
        ops.downblock_fused(x, p1, b, pd, b, p2, b, out=out)
    t2, t1 = timed(two), timed(one)
    gf = 2.0 * n * oh * ow * 64 * (576 + 64 + 576) / 1e9
    print('%2d x %3d x %3d: two launches %6.1f us, fused %6.1f us  (%.0f TF, in %.0f MB -> %.2f TB/s)' % (
        n, h, w, t2, t1, gf / t1 / 1e3 * 1e0, n * h * w * 128 / 1e6, (n * h * w * 128 + n * oh * ow * 128) / t1 / 1e6))
