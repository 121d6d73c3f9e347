"""Shader clock / socket power of the MI355X while (a) a pure MFMA loop, (b) lfd_fasterblock_fused_f16 (k_block64: 8 x 16 tiles; k_block64_rows: row streaming), lfd_downblock_fused_f16 (k_down64) and
(c) the headline bench step run back to back for a few seconds each: rocm-smi polled from a side thread plus the in-kernel
s_memtime / s_memrealtime ratio where the kernel has stamps.  Evidence for which ceiling the MFMA fractions are quoted
against ("commit an sclk / power trace next to the kernel stats") -> gpurun_out/power_trace.json."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

samples = []
phase = ['idle']
stop = [False]


def poll():
    while not stop[0]:
        t = time.time()
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showuse', '--json'], capture_output=True, text=True,
                                 timeout=10).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            rec = {'t': round(t, 2), 'phase': phase[0]}
            for k, v in card.items():
                kl = k.lower()
                if 'power' in kl or 'sclk' in kl or 'mclk' in kl or 'gpu use' in kl:
                    rec[k] = v
            samples.append(rec)
        except Exception as e:      # keep polling
            samples.append({'t': round(t, 2), 'phase': phase[0], 'error': repr(e)})
        time.sleep(0.1)


def run_for(seconds, fn, sync_every=50):
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(sync_every):
            fn()
        torch.cuda.synchronize()
        n += sync_every
    return n, time.time() - t0


def main():
    from lfd_amd import ops
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    time.sleep(1.0)
    res = {}
    # (a) pure MFMA: tools/ub/mfma.hip built on the box
    exe = '/tmp/ub_mfma'
    phase[0] = 'build'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', os.path.join(ROOT, 'tools', 'ub', 'mfma_long.hip'), '-o', exe], check=True)
    phase[0] = 'pure_mfma'
    out = subprocess.run([exe], capture_output=True, text=True).stdout
    res['pure_mfma'] = out.strip().splitlines()
    phase[0] = 'idle2'
    time.sleep(1.0)
    # (b) k_block64 at 32 x 135 x 240
    g = torch.Generator().manual_seed(0)
    w1 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
    w2 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
    b1, b2 = torch.randn(64, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
    x = (torch.randn(32, 135, 240, 64, generator=g) * 0.5).half().cuda()
    y = torch.empty_like(x)
    gf = 2 * 2.0 * 32 * 135 * 240 * 64 * 64 * 9 / 1e9
    # the 8 x 16-tile kernel in a child process (LFD_BLOCK_ROWS is read once per process), the row-streaming kernel here
    phase[0] = 'k_block64'
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child_tiles'], env=dict(os.environ, LFD_BLOCK_ROWS='0'),
                       capture_output=True, text=True)
    try:
        res['k_block64'] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res['k_block64'] = {'error': r.stderr[-500:]}
    phase[0] = 'idle2b'
    time.sleep(1.0)
    phase[0] = 'k_block64_rows'
    n, dt = run_for(4.0, lambda: ops.fasterblock_fused(x, w1, b1, w2, b2, out=y))
    res['k_block64_rows'] = dict(launches=n, us_per_launch=round(dt / n * 1e6, 2), tflops=round(gf * n / dt / 1e3, 1))
    phase[0] = 'idle2c'
    time.sleep(1.0)
    # the fused downsample block at 8 x 270 x 480 -> 135 x 240 (four rotating inputs: cold reads)
    pd = ops.pack_conv_weight((torch.randn(64, 64, 1, 1, generator=g) / 8)).cuda()
    xd = [(torch.randn(8, 270, 480, 64, generator=g) * 0.5).half().cuda() for _ in range(4)]
    yd = torch.empty(8, 135, 240, 64, dtype=torch.float16, device='cuda')
    kk = [0]
    def down():
        kk[0] += 1
        ops.downblock_fused(xd[kk[0] & 3], w1, b1, pd, b2, w2, b2, out=yd)
    phase[0] = 'k_down64'
    n, dt = run_for(4.0, down)
    gfd = 2.0 * 8 * 135 * 240 * 64 * 64 * (9 + 1 + 9) / 1e9
    res['k_down64'] = dict(launches=n, us_per_launch=round(dt / n * 1e6, 2), tflops=round(gfd * n / dt / 1e3, 1),
                           tb_per_s=round((8 * 270 * 480 + 8 * 135 * 240) * 128 * n / dt / 1e12, 2))
    phase[0] = 'idle3'
    time.sleep(1.0)
    # (c) the headline step (WIDERFACE_LFD_S 8 x 1080p, forward + decode + NMS, one HIP graph, serial replay)
    from lfd_amd import configs
    m = configs.build_model('WIDERFACE_LFD_S')
    configs.perturb_weights(m)
    m.eval().cuda()
    m.use_graph = True
    xs = (torch.rand(8, 1080, 1920, 3, device='cuda') * 2 - 1).half()
    meta = torch.tensor([[1920., 1080., 1.0]] * 8, device='cuda')
    with torch.no_grad():
        cls, _ = m.forward_resident(xs)
        m._classification_threshold = float(torch.quantile(cls.float().sigmoid().reshape(8, -1)[0], 1.0 - 256 / cls.shape[1]))
        m.detect_resident(xs, meta)
        torch.cuda.synchronize()
        phase[0] = 'bench_step'
        n, dt = run_for(4.0, lambda: m.detect_resident(xs, meta), sync_every=20)
    res['bench_step'] = dict(steps=n, ms_per_step=round(dt / n * 1e3, 4), tflops=round(348.8 * n / dt / 1e3, 1))
    phase[0] = 'idle4'
    time.sleep(0.5)
    stop[0] = True
    th.join(timeout=5)
    res['samples'] = samples
    # per-phase summary of every numeric-looking field
    summ = {}
    for s in samples:
        for k, v in s.items():
            if k in ('t', 'phase', 'error'):
                continue
            try:
                f = float(str(v).strip('()MhzW% ').split()[0].replace('Mhz', ''))
            except Exception:
                continue
            summ.setdefault(s['phase'], {}).setdefault(k, []).append(f)
    res['summary'] = {p: {k: dict(n=len(v), min=min(v), max=max(v), mean=round(sum(v) / len(v), 1)) for k, v in d.items()} for p, d in summ.items()}
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'power_trace.json'), 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != 'samples'}, indent=1))


def child_tiles():
    from lfd_amd import ops
    g = torch.Generator().manual_seed(0)
    w1 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
    w2 = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / 24)).cuda()
    b1, b2 = torch.randn(64, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
    x = (torch.randn(32, 135, 240, 64, generator=g) * 0.5).half().cuda()
    y = torch.empty_like(x)
    n, dt = run_for(4.0, lambda: ops.fasterblock_fused(x, w1, b1, w2, b2, out=y))
    gf = 2 * 2.0 * 32 * 135 * 240 * 64 * 64 * 9 / 1e9
    print(json.dumps(dict(launches=n, us_per_launch=round(dt / n * 1e6, 2), tflops=round(gf * n / dt / 1e3, 1))))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child_tiles':
        child_tiles()
    else:
        main()
