"""Per-phase cycle stamps of k_pl_stem2x (csrc/planes_stem2x.hip; producer phases in csrc/planes_impl.h) for wave 0 of workgroup 0:
    tools/ab_build.sh scratch/alt/liblfd_hip_plt.so -DLFD_PL_TIMING; LFD_HIP_LIB=$PWD/scratch/alt/liblfd_hip_plt.so python tools/timing/stem2x_phases.py"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import ops, engine_p2
from lfd_amd._lib import check, lib, ptr, stream_ptr
L = lib()
has_t = hasattr(L, 'lfd_debug_pl_stem2x_timing')
if has_t: L.lfd_debug_pl_stem2x_timing.argtypes = [C.c_void_p]
dev = torch.device('cuda')
n, h, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 1080, 1920
g = torch.Generator().manual_seed(0)
x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda()
ws = [torch.randn(64, 3, 3, 3, generator=g) * 0.3, torch.randn(64, 64, 1, 1, generator=g) * 0.1, torch.randn(64, 64, 3, 3, generator=g) * 0.04,
      torch.randn(64, 64, 1, 1, generator=g) * 0.1]
b = torch.zeros(128, device=dev)
keep = [engine_p2.pack_planes_stem2x_weight(ws[0], torch.zeros(64)).cuda(), engine_p2.pack_planes_stem2x_tail_weight(ws[1]).cuda()] + [engine_p2.pack_planes_weight(t).cuda() for t in ws[2:]]
out = torch.empty((2, n, 270, 480, 64), dtype=torch.float16, device=dev)
z = ops.zero_line(dev)
def run():
    check(L.lfd_pl_stem2x(ptr(x), 1, n, h, w, ptr(keep[0]), ptr(keep[1]), ptr(b), ptr(keep[2]), ptr(b), ptr(keep[3]), ptr(b), ptr(out),
                          out[0].numel(), ptr(z), stream_ptr()), 'stem2x')
for _ in range(5): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print('stem2x %dx%dx%d: %.1f us' % (n, h, w, e0.elapsed_time(e1) * 100))
if has_t:
    buf = (C.c_ulonglong * 128)(); L.lfd_debug_pl_stem2x_timing(buf)
    for it in range(1, 5):
        v = [buf[it * 16 + i] for i in range(16)]
        nxt = buf[(it + 1) * 16]
        print('   tile %d: top-wait+barrier %d  frame-dma %d  produce %d [round 2: conv0(r+1) issue %d  1x1 issue %d  to_frag %d  D %d  (exchange + rest: %d per round)]  ..kloop-start %d  '
              'kloop %d  tail %d  barrier %d  stage %d  wait+barrier %d  copy-out %d | total %d  next-gap %d'
              % (it, v[10] - v[0], v[11] - v[10], v[12] - v[11], v[2] - v[1], v[13] - v[2], v[14] - v[13], v[15] - v[14], (v[12] - v[11]) // 5 - (v[15] - v[1]), v[3] - v[12], v[4] - v[3], v[5] - v[4],
                 v[6] - v[5], v[7] - v[6], v[8] - v[7], v[9] - v[8], v[9] - v[0], nxt - v[9]))
