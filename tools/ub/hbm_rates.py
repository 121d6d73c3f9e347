import torch, json
def t(f, reps=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (178, 357, 1062):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.float32, device='cuda'); b = torch.empty_like(a)
    gb = n * 4 / 1e9
    r = dict(MB=mb)
    us = t(lambda: a.fill_(1.0)); r['fill_us'] = round(us, 1); r['write_TBps'] = round(gb / us * 1e3, 2)
    us = t(lambda: b.copy_(a)); r['copy_us'] = round(us, 1); r['copy_rw_TBps'] = round(2 * gb / us * 1e3, 2)
    us = t(lambda: a.sum()); r['read_us'] = round(us, 1); r['read_TBps'] = round(gb / us * 1e3, 2)
    print(json.dumps(r))
