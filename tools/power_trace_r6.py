"""Socket power and shader clock of the MI355X per KERNEL CLASS of both precision modes and for the whole steps (round 6;
VERDICT r5 next-round item 3: "a tools/power_trace.py record under profiles/ showing it at the socket power limit with the
clock it holds").  Every launch unit of bench.py's two breakdowns runs back to back for SECONDS seconds while rocm-smi is
polled by a child process; then the whole steps (HIP graph, one and two batches in flight).
    python tools/power_trace_r6.py [seconds per unit] [fp16|precise|steps ...]   -> gpurun_out/power_trace_r6.json"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

# rocm-smi is polled by a CHILD PROCESS (round 6: a polling thread in this interpreter shares the GIL with the launch loop and
# made the host the bottleneck of the short kernels and of the fp16 step); samples carry wall-clock time stamps, phases are
# (name, t0, t1) windows recorded here
_POLLER = r"""
import json, subprocess, sys, time
out = open(sys.argv[1], 'w')
def num(v):
    try:
        return float(str(v).strip('()MhzW% ').split()[0].replace('Mhz', ''))
    except Exception:
        return None
while True:
    t = time.time()
    try:
        d = json.loads(subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10).stdout)
        card = d[sorted(d)[0]]
        rec = {'t': t}
        for k, v in card.items():
            kl = k.lower()
            if 'power' in kl:
                rec['power_w'] = num(v)
            elif 'sclk clock speed' in kl:
                rec['sclk_mhz'] = num(v)
        out.write(json.dumps(rec) + '\n'); out.flush()
    except Exception as e:
        pass
    time.sleep(0.02)
"""
windows = {}
_poll_file = '/tmp/lfd_power_samples_%d.jsonl' % os.getpid()
_poll_proc = None


def start_poller():
    global _poll_proc
    src = '/tmp/lfd_power_poller_%d.py' % os.getpid()
    open(src, 'w').write(_POLLER)
    _poll_proc = subprocess.Popen([sys.executable, src, _poll_file])


def stop_poller():
    if _poll_proc is not None:
        _poll_proc.terminate()


def _samples():
    try:
        return [json.loads(l) for l in open(_poll_file) if l.strip()]
    except Exception:
        return []


def summarize(name):
    t0, t1 = windows[name]
    rows = [r for r in _samples() if t0 + 0.25 * (t1 - t0) <= r['t'] <= t1]        # (the first quarter of a phase is the ramp)
    if not rows:
        return {}
    pw = [r['power_w'] for r in rows if r.get('power_w') is not None]
    ck = [r['sclk_mhz'] for r in rows if r.get('sclk_mhz') is not None]
    return dict(samples=len(rows), power_w_mean=round(float(np.mean(pw)), 0) if pw else None, power_w_max=max(pw) if pw else None,
                sclk_mhz_mean=round(float(np.mean(ck)), 0) if ck else None, sclk_mhz_min=min(ck) if ck else None)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].replace('.', '').isdigit() else 2.0
    what = [a for a in sys.argv[1:] if not a.replace('.', '').isdigit()] or ['fp16', 'precise', 'steps']
    import bench
    from lfd_amd import configs, engine, _lib
    start_poller()
    dev = torch.device('cuda')
    m = configs.build_model('WIDERFACE_LFD_S')
    configs.perturb_weights(m)
    m.eval().cuda()
    m.use_graph = True
    m.max_candidates = 8192
    m._nms_cfg = dict(type='nms', iou_thr=0.4)
    xs = (torch.rand(bench.NBUF, 8, 1080, 1920, 3, device=dev) * 2 - 1).half()      # bench.py's rotation of resident frame buffers
    x = xs[0]
    meta = torch.tensor([[1920., 1080., 1.0]] * 8, device=dev)
    res = {'seconds_per_unit': seconds, 'idle': None, 'units': [], 'steps': []}
    t_idle = time.time()
    time.sleep(1.5)
    windows['idle'] = (t_idle, time.time())
    res['idle'] = summarize('idle')
    ctr = [0]

    def timer(fn, label=''):
        ctr[0] += 1
        name = 'u%d' % ctr[0]
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        while time.time() - t0 < seconds:
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            n += 200
        dt = time.time() - t0
        windows[name] = (t0, t0 + dt)
        time.sleep(0.1)
        us = dt / n * 1e6
        row = dict(unit=label, us_per_launch_back_to_back=round(us, 2))
        row.update(summarize(name))
        res['units'].append(row)
        print(json.dumps(row), flush=True)
        time.sleep(0.3)
        return us

    with torch.no_grad():
        cls, _ = m.forward_resident(x)
        m._classification_threshold = float(torch.quantile(cls.float().sigmoid().reshape(8, -1)[0], 1.0 - 256 / cls.shape[1]))
        if 'fp16' in what:
            fmt, n, h, w = engine._input_format(x)
            plan = engine.get_plan(m, m._backbone, m._neck, m._head, dev)
            st = plan.state_for(n, h, w)
            res['units'].append(dict(unit="==== LFD.precision = 'fp16' launch units"))
            bench.kernel_breakdown(m, plan, st, x, fmt, timer=timer)
        if 'precise' in what:
            m.precision = 'fp32_storage'
            res['units'].append(dict(unit="==== LFD.precision = 'fp32_storage' launch units (PL_C3 = %d)" % _lib.tune('PL_C3')))
            bench.precise_breakdown(m, x, dev, timer=timer)
            m.precision = 'fp16'
        if 'steps' in what:
            # the whole steps, measured by bench.py's own timed_region() (graphs captured for the stream set-up they replay on, four
            # frame buffers in rotation, two batches in flight | serial): its >= 3 s gap-free replays are the power windows
            import argparse
            args = argparse.Namespace(steps=20, warmup=5, clock_warmup_s=0.3, sustained_s=max(seconds, 3.0))
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            for mode in ('fp16', 'fp32_storage'):
                m.precision = mode
                r = bench.timed_region(m, xs, meta, args, 2, streams, dev, sync_ranks=False)
                for name, depth in (('serial', 1), ('pipelined', 2)):
                    su = r['sustained'][name]
                    windows['step_%s_%s' % (mode, name)] = tuple(su['wall_clock_window'])
                    row = dict(step="%s, %d batch(es) in flight, HIP graph replays for %.1f s without host synchronisation" % (mode, depth, su['seconds']),
                               ms_per_step=su['ms_per_step'], images_per_s=su['images_per_s'],
                               mfma_tflops_issued=round((3 if mode != 'fp16' else 1) * 348.8 / su['ms_per_step'], 1))
                    row.update(summarize('step_%s_%s' % (mode, name)))
                    res['steps'].append(row)
                    print(json.dumps(row), flush=True)
            m.precision = 'fp16'
    stop_poller()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    tag = os.environ.get('LFD_POWER_TAG', '')
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'power_trace_r6%s.json' % tag), 'w'), indent=1)


if __name__ == '__main__':
    main()
