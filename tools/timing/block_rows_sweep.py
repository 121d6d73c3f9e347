"""k_block64 (LFD_BLOCK_ROWS=0) vs k_block64_rows (=1): per-shape time of lfd_fasterblock_fused_f16 from HIP graphs of 20
chained launches over 4 rotating inputs.  The mode is read once per process: this script re-executes itself per mode."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(1, 135, 240), (2, 135, 240), (4, 135, 240), (8, 135, 240), (16, 135, 240), (32, 135, 240), (8, 68, 120), (32, 68, 120),
          (8, 34, 60), (1, 270, 480), (1, 540, 960), (4, 180, 320), (32, 160, 160), (8, 100, 100)]
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    for p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')): sys.path.insert(0, p)
    import torch
    from lfd_amd import ops
    g = torch.Generator().manual_seed(0)
    p1 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    p2 = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
    b = (torch.randn(64, generator=g) * 0.1).cuda()
    out = {}
    for (n, h, w) in SHAPES:
        xs = [(torch.randn(n, h, w, 64, generator=g) * 0.5).half().cuda() for _ in range(4)]
        y = torch.empty_like(xs[0])
        def fn(i): ops.fasterblock_fused(xs[i % 4], p1, b, p2, b, out=y)
        fn(0); torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for i in range(20): fn(i)
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        out['%dx%dx%d' % (n, h, w)] = round(min(ts), 2)
    print(json.dumps(out))
else:
    res = {}
    for mode in ('0', '1'):
        env = dict(os.environ, LFD_BLOCK_ROWS=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res['0']:
        print('%-14s tiles %7.2f us   rows %7.2f us   %+5.1f %%' % (k, res['0'][k], res['1'][k], 100.0 * (res['1'][k] / res['0'][k] - 1)))
