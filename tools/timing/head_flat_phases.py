"""Phase stamps + launch times of k_pl_head (csrc/planes_head.hip) on one pyramid level, modes 0 / 1 / 2, next to the generic
multi-level conv (lfd_pl_conv2d_levels) on the same level.   LFD_HIP_LIB=<-DLFD_PL_TIMING build> python tools/timing/head_flat_phases.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import ops, _lib, engine_p2
from lfd_amd._lib import check, lib, ptr, stream_ptr
L = lib()
dev = torch.device('cuda')
z = ops.zero_line(dev)
has_t = hasattr(L, 'lfd_debug_pl_head_timing')
if has_t: L.lfd_debug_pl_head_timing.argtypes = [C.c_void_p]
n, h, w = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 135, 240)))
p = h * w
g = torch.Generator().manual_seed(0)
xp = engine_p2.to_planes(torch.randn(n, p, 1, 64, generator=g)).to(dev)
pk = lambda co, ci: engine_p2.pack_planes_weight(torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).to(dev)
w0, w1, w2, w3 = pk(128, 64), pk(128, 128), pk(128, 128), pk(5, 128)
b = torch.zeros(128, device=dev)
ga, be = torch.ones(128, device=dev), torch.zeros(128, device=dev)
t1, t2 = torch.empty(n, p, 128, device=dev), torch.empty(n, p, 128, device=dev)
gs = torch.zeros((2, _lib.PL_GN_REPLICAS, n, 16, 2), dtype=torch.int64, device=dev)
cls, reg = torch.empty(n, p, 1, device=dev), torch.empty(n, p, 4, device=dev)
lv = [(_lib.PlHeadLevel * 1)() for _ in range(3)]
a, bb, c = lv[0][0], lv[1][0], lv[2][0]
a.in_, a.out, a.w0, a.b0, a.w1, a.b1, a.gn_sums, a.in_plane_halfs, a.pixels = xp.data_ptr(), t1.data_ptr(), w0.data_ptr(), b.data_ptr(), w1.data_ptr(), b.data_ptr(), gs[0].data_ptr(), xp[0].numel(), p
bb.in_, bb.out, bb.w0, bb.b0, bb.gn_sums, bb.gn_in_sums, bb.gn_in_gamma, bb.gn_in_beta, bb.pixels = t1.data_ptr(), t2.data_ptr(), w2.data_ptr(), b.data_ptr(), gs[1].data_ptr(), gs[0].data_ptr(), ga.data_ptr(), be.data_ptr(), p
c.in_, c.w0, c.b0, c.gn_in_sums, c.gn_in_gamma, c.gn_in_beta, c.f_out0, c.f_out1, c.pixels = t2.data_ptr(), w3.data_ptr(), b.data_ptr(), gs[1].data_ptr(), ga.data_ptr(), be.data_ptr(), cls.data_ptr(), reg.data_ptr(), p
NAMES = ['top->wait', 'barrier', 'level+gn-transform', 'barrier2', 'dma-issue', 'contract', 'epilogue']
def run(mode):
    d = _lib.PlHeadDesc()
    d.mode, d.n, d.cin, d.relu0, d.gn_in_eps, d.f_c0, d.f_c1, d.f_image_stride0, d.f_image_stride1 = mode, n, 64, 1, 1e-5, 1, 4, p, p * 4
    check(L.lfd_pl_head_levels(C.byref(d), lv[mode], 1, ptr(z), stream_ptr()), 'head')
t0 = time.time()
while time.time() - t0 < 0.3:
    run(0); torch.cuda.synchronize()
for mode in (0, 1, 2):
    gs.zero_(); run(0); run(1); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(mode)
    e1.record(); torch.cuda.synchronize()
    mb = {0: n * p * (256 + 512), 1: n * p * 1024, 2: n * p * (512 + 20)}[mode] / 1e6
    us = e0.elapsed_time(e1) * 50
    print('mode %d  %dx%dx%d: %.1f us  (%.0f MB -> %.2f TB/s)' % (mode, n, h, w, us, mb, mb / us / 1e6 * 1e6 / 1e6))
    if has_t:
        buf = (C.c_ulonglong * 136)(); L.lfd_debug_pl_head_timing(buf)
        for it in range(1, 5):
            v = [buf[it * 16 + i] for i in range(8)]
            if mode == 0:
                print('   tile %d: wait %d barrier %d | compute(level + neck + mid + conv1) %d | epilogue %d | total %d  gap-to-next %d' % (
                    it, v[1] - v[0], v[2] - v[1], v[6] - v[2], v[7] - v[6], v[7] - v[0], buf[(it + 1) * 16] - v[7]))
            else:
                print('   tile %d: ' % it + '  '.join('%s %d' % (NAMES[i], v[i + 1] - v[i]) for i in range(7)) + ' | total %d' % (v[7] - v[0]))
