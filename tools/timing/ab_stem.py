"""same-process A/B of two liblfd_hip.so builds on lfd_stem_faster_fused_f16 (warm clocks, interleaved rounds)"""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
from lfd_amd import configs, engine
libs = [C.CDLL(p) for p in sys.argv[1:3]]
for l in libs:
    l.lfd_stem_faster_fused_f16.argtypes = [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 10
m = configs.build_model('WIDERFACE_LFD_S'); configs.perturb_weights(m); m.eval().cuda()
plan = engine.get_plan(m, m._backbone, m._neck, m._head, torch.device('cuda', 0))
c0, w1, b1, w2, b2, w3, b3, w4, b4, dst = plan.stem_fused
for (n, h, w) in ((8, 1080, 1920), (1, 1080, 1920), (4, 720, 1280)):
    x = (torch.rand(n, h, w, 3, device='cuda') * 2 - 1).half()
    ys = [torch.empty((n, (h + 3) // 4, (w + 3) // 4, 64), dtype=torch.float16, device='cuda') for _ in range(2)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(i):
        rc = libs[i].lfd_stem_faster_fused_f16(x.data_ptr(), 1, n, h, w, c0, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                               w3.data_ptr(), b3.data_ptr(), w4.data_ptr(), b4.data_ptr(), ys[i].data_ptr(), st)
        assert rc == 0, rc
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(10): run(0); run(1)
        torch.cuda.synchronize()
    res = [[], []]
    for _ in range(7):
        for i in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(i)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 50)
    med = [sorted(r)[3] for r in res]
    print(json.dumps(dict(shape=[n, h, w], a_us=round(med[0], 2), b_us=round(med[1], 2), b_over_a=round(med[1] / med[0], 3), identical=bool(torch.equal(ys[0], ys[1])))))
