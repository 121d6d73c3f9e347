"""same-process A/B of two builds on lfd_conv2d_downsample_nhwc_f16 (3x3 s2 64->64 + 1x1 s2 branch) in graph-captured chains"""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
from lfd_amd import ops, _lib
libs = [C.CDLL(p) for p in sys.argv[1:3]]
for l in libs:
    l.lfd_conv2d_downsample_nhwc_f16.argtypes = [C.POINTER(_lib.ConvDesc)] + [C.c_void_p] * 9
g = torch.Generator().manual_seed(0)
w = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
wd = ops.pack_conv_weight((torch.randn(64, 64, 1, 1, generator=g) / 8)).cuda()
b, bd = (torch.randn(64, generator=g) * 0.1).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
z = ops.zero_line(torch.device('cuda', 0))
for (n, h, wd_) in ((8, 270, 480), (8, 135, 240), (1, 270, 480), (4, 360, 640)):
    x = (torch.randn(n, h, wd_, 64, generator=g) * 0.5).half().cuda()
    oh, ow = (h + 1) // 2, (wd_ + 1) // 2
    ys = [[torch.empty(n, oh, ow, 64, dtype=torch.float16, device='cuda') for _ in range(2)] for _ in range(2)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = _lib.ConvDesc(n, h, wd_, 64, 64, 3, 2, 1, 0, 0)
    def run(i):
        rc = libs[i].lfd_conv2d_downsample_nhwc_f16(C.byref(d), x.data_ptr(), ys[i][0].data_ptr(), w.data_ptr(), b.data_ptr(), wd.data_ptr(), bd.data_ptr(), ys[i][1].data_ptr(), z.data_ptr(), st)
        assert rc == 0, rc
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(20): run(0); run(1)
        torch.cuda.synchronize()
    res = [[], []]
    for _ in range(7):
        for i in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run(i)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 1e3 / 30)
    med = [sorted(r)[3] for r in res]
    print(json.dumps(dict(shape=[n, h, wd_], a_us=round(med[0], 2), b_us=round(med[1], 2), b_over_a=round(med[1] / med[0], 3),
                          identical=bool(torch.equal(ys[0][0], ys[1][0]) and torch.equal(ys[0][1], ys[1][1])))))
