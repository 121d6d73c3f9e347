"""GPU time of a 128-channel FasterBlock at 17x30 inside a HIP graph: the fused launch (csrc/block128.hip) against its two
split-K conv launches (csrc/conv_small.hip); chains of 10 blocks, so launch boundaries are part of the number"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import ops
g = torch.Generator().manual_seed(0)
w1 = ops.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) / 34)).cuda()
w2 = ops.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) / 34)).cuda()
b1 = (torch.randn(128, generator=g) * 0.1).cuda()
b2 = (torch.randn(128, generator=g) * 0.1).cuda()
shapes = [(1, 17, 30), (8, 17, 30), (8, 12, 20), (1, 34, 60), (16, 23, 40)]
for n, h, w in shapes:
    x = (torch.randn(n, h, w, 128, generator=g) * 0.5).half().cuda()
    bufs = [torch.empty_like(x) for _ in range(3)]

    def chain(fused, k=10):
        src = x
        for i in range(k):
            dst = bufs[i & 1]
            if fused:
                ops.fasterblock128_fused(src, w1, b1, w2, b2, out=dst)
            else:
                ops.conv2d_nhwc(src, w1, b1, 128, 128, 3, 1, True, out=bufs[2])
                ops.conv2d_nhwc(bufs[2], w2, b2, 128, 128, 3, 1, True, residual=src, out=dst)
            src = dst
        return src
    res = {}
    outs = {}
    for fused in (0, 1):
        outs[fused] = chain(fused).clone(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            chain(fused)
        for _ in range(20): gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 10)
        ts.sort()
        res['fused' if fused else 'two_launches'] = round(ts[len(ts) // 2], 2)
    print(json.dumps(dict(shape=[n, h, w], us_per_block=res, identical=bool(torch.equal(outs[0], outs[1])))), flush=True)
