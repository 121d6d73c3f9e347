"""GPU time per launch of the 128->128 3x3 conv at 17x30 inside a HIP graph (eager timing is host-bound at ~8 us per call)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import ops
g = torch.Generator().manual_seed(0)
w = ops.pack_conv_weight((torch.randn(128, 128, 3, 3, generator=g) / 34)).cuda()
b = (torch.randn(128, generator=g) * 0.1).cuda()
for n in (1, 8):
    x = (torch.randn(n, 17, 30, 128, generator=g) * 0.5).half().cuda()
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    def chain(k=20):
        src = x
        for i in range(k):
            ops.conv2d_nhwc(src, w, b, 128, 128, 3, 1, True, residual=x, out=bufs[i & 1])
            src = bufs[i & 1]
    chain(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    ts.sort()
    print(json.dumps(dict(splitk=os.environ.get('LFD_CONV128_SPLITK', '1'), n=n, us_per_launch_in_graph=round(ts[len(ts) // 2], 2), min=round(ts[0], 2))))
