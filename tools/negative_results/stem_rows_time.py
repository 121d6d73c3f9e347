"""k_stem2x (LFD_STEM_ROWS=0) vs k_stem_rows (=1): time of lfd_stem_faster_fused_f16 per shape from HIP graphs of 10 launches over 4
rotating frame batches.  The mode is read once per process: this script re-executes itself per mode."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(8, 1080, 1920), (1, 1080, 1920), (2, 1080, 1920), (1, 2160, 3840), (4, 720, 1280), (8, 640, 640)]
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    for p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')): sys.path.insert(0, p)
    import torch
    from lfd_amd import engine, ops
    from lfd_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(0)
    c = 64
    ws = [(torch.randn(c, 3, 3, 3, generator=g) * 0.2), (torch.randn(c, c, 1, 1, generator=g) / 8), (torch.randn(c, c, 3, 3, generator=g) / 24),
          (torch.randn(c, c, 1, 1, generator=g) / 8)]
    bs = [(torch.randn(c, generator=g) * 0.1).cuda() for _ in range(4)]
    pk = [engine.pack_stem_weight(ws[0]).cuda()] + [ops.pack_conv_weight(t).cuda() for t in ws[1:]]
    out = {}
    for (n, h, w) in SHAPES:
        xs = [(torch.rand(n, h, w, 3, generator=g) * 2 - 1).half().cuda() for _ in range(4)]
        h2, w2 = ((h + 1) // 2 + 1) // 2, ((w + 1) // 2 + 1) // 2
        y = torch.empty(n, h2, w2, c, dtype=torch.float16, device='cuda')
        def fn(i):
            check(lib().lfd_stem_faster_fused_f16(ptr(xs[i % 4]), 1, n, h, w, c, ptr(pk[0]), ptr(bs[0]), ptr(pk[1]), ptr(bs[1]), ptr(pk[2]),
                                                  ptr(bs[2]), ptr(pk[3]), ptr(bs[3]), ptr(y), stream_ptr()), 'stem')
        fn(0); torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for i in range(10): fn(i)
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 10)
        out['%dx%dx%d' % (n, h, w)] = round(min(ts), 2)
    print(json.dumps(out))
else:
    res = {}
    for mode in ('0', '1'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, LFD_STEM_ROWS=mode), capture_output=True, text=True)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res['0']:
        print('%-14s k_stem2x %7.2f us   rows %7.2f us   %+5.1f %%' % (k, res['0'][k], res['1'][k], 100.0 * (res['1'][k] / res['0'][k] - 1)))
