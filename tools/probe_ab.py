"""same-process A/B of two liblfd_hip.so builds on lfd_fasterblock_fused_f16 (warm clocks, interleaved rounds)"""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
from lfd_amd import ops
libs = [C.CDLL(p) for p in sys.argv[1:3]]
for l in libs:
    l.lfd_fasterblock_fused_f16.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 8
g = torch.Generator().manual_seed(0)
w1 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
w2 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
b1, b2 = torch.randn(64, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
z = ops.zero_line(torch.device('cuda', 0))
for (n, h, w) in ((8, 135, 240), (32, 135, 240), (8, 68, 120), (4, 180, 320)):
    x = (torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda()
    ys = [torch.empty_like(x), torch.empty_like(x)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(i):
        rc = libs[i].lfd_fasterblock_fused_f16(n, h, w, x.data_ptr(), ys[i].data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), z.data_ptr(), st)
        assert rc == 0
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(20): run(0); run(1)
        torch.cuda.synchronize()
    res = [[], []]
    for _ in range(7):
        for i in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run(i)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 20)
    med = [sorted(r)[3] for r in res]
    print(json.dumps(dict(shape=[n, h, w], a_us=round(med[0], 2), b_us=round(med[1], 2), b_over_a=round(med[1] / med[0], 3), identical=bool(torch.equal(ys[0], ys[1])))))
