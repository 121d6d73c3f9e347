"""lfd_fasterblock_fused_f16 at 32 x 135 x 240 for 4 s (the chip settles at its power cap): us per launch of the library named by
LFD_HIP_LIB / the kernel picked by LFD_BLOCK_ROWS."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')): sys.path.insert(0, p)
import torch
from lfd_amd import ops, _lib
g = torch.Generator().manual_seed(0)
w1 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
w2 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
b = (torch.randn(64, generator=g) * 0.1).cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = (torch.randn(n, 135, 240, 64, generator=g) * 0.5).half().cuda()
y = torch.empty_like(x)
t0 = time.time(); k = 0
while time.time() - t0 < 4.0:
    for _ in range(50): ops.fasterblock_fused(x, w1, b, w2, b, out=y)
    torch.cuda.synchronize(); k += 50
    if time.time() - t0 < 2.0: k2, t2 = k, time.time()
dt = time.time() - t2
print(json.dumps(dict(lib=os.path.basename(_lib.LIB_PATH), rows=os.environ.get('LFD_BLOCK_ROWS'), n=n, us_per_launch_last_2s=round(dt / (k - k2) * 1e6, 2))))
