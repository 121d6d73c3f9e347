"""HBM write / read / copy rates of this box with plain PyTorch kernels (fill_, sum, copy_) on 1.06 GB -- the size of the first
stem pair's output planes at 8 x 1080p: what "write-bound" means for that kernel."""
import torch, json
n = 8 * 540 * 960 * 64 * 2   # halfs: both planes
a = torch.empty(n, dtype=torch.float16, device='cuda'); b = torch.empty_like(a)
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
gb = n * 2 / 1e9
r = dict(gigabytes=round(gb, 3))
us = t(lambda: a.fill_(1.0)); r['fill_us'] = round(us, 1); r['fill_TBps'] = round(gb / us * 1e3 / 1e3, 2)
us = t(lambda: b.copy_(a)); r['copy_us'] = round(us, 1); r['copy_TBps_rw'] = round(2 * gb / us * 1e3 / 1e3, 2)
af = a.view(torch.float32)
us = t(lambda: af.sum()); r['read_us'] = round(us, 1); r['read_TBps'] = round(gb / us * 1e3 / 1e3, 2)
print(json.dumps(r))
