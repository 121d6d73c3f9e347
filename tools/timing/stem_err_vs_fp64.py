import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch, torch.nn.functional as F
from lfd_amd import ops, engine
from lfd_amd._lib import check, lib, ptr, stream_ptr
g = torch.Generator().manual_seed(2); c=64
n,h,w = 1,540,960
ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5),
      (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5), (torch.randn(c, c, 1, 1, generator=g) / c ** 0.5)]
ws = [t.half().float() for t in ws]
bs = [torch.randn(c, generator=g) * 0.1 for _ in range(4)]
xf = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).half().float()
y = xf.double().cuda()
for wt, b, s, p in zip(ws, bs, (2, 1, 2, 1), (1, 0, 1, 0)):
    y = F.conv2d(y, wt.double().cuda(), b.double().cuda(), stride=s, padding=p).relu().float().half().double()
packed = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
bg = [b.cuda() for b in bs]
out = torch.empty((n, y.shape[2], y.shape[3], c), dtype=torch.float16).cuda()
xin = xf.permute(0, 2, 3, 1).contiguous().half().cuda()
check(lib().lfd_stem_faster_fused_f16(ptr(xin), 1, n, h, w, c, ptr(packed[0]), ptr(bg[0]), ptr(packed[1]), ptr(bg[1]),
      ptr(packed[2]), ptr(bg[2]), ptr(packed[3]), ptr(bg[3]), ptr(out), stream_ptr()), 'fused stem')
torch.cuda.synchronize()
got = out.double().permute(0, 3, 1, 2)
e = (got - y).abs()
print(os.environ.get('LFD_HIP_LIB','cur')[-16:], 'mean abs err %.4e  max %.4e  frac nonzero %.4f  rel-L2 %.4e' % (float(e.mean()), float(e.max()), float((e>0).double().mean()), float((e.pow(2).sum()/y.pow(2).sum()).sqrt())))
