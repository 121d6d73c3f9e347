"""Sibling meta-architectures on the device: forward (eager launches vs one HIP graph) and get_results
(lfd_detect_batched_ex) per batch, HIP-event timed, with the CPU restatement (oracle/sibling_oracle.py, fp32 eager PyTorch
on the host cores) timed beside it.  One JSON line per configuration.

    python tools/bench_siblings.py [--n 8] [--h 720] [--w 1280] [--reps 30] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
import torch
from lfd_amd import configs


def ev_time(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts)), float(np.percentile(ts, 95))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--h', type=int, default=720)
    ap.add_argument('--w', type=int, default=1280)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--no-cpu', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    for name in sorted(configs.SIBLINGS):
        spec = configs.SIBLINGS[name]
        model = configs.build_sibling_model(name, seed=1).eval().to(dev)
        x = (torch.rand(a.n, 3, a.h, a.w, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)
        meta = torch.tensor([[float(a.w), float(a.h), 1.0]] * a.n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            model.use_graph = False
            outs = [o.clone() for o in model(x)]
            torch.cuda.synchronize()
            eager = ev_time(lambda: model(x), a.reps)
            model.use_graph = True
            fwd = (lambda: model.forward_resident(x)) if hasattr(model, 'forward_resident') else (lambda: model(x))
            fwd()
            torch.cuda.synchronize()
            graph = ev_time(fwd, a.reps)
            if len(outs) == 3:
                sc = outs[0].sigmoid() * outs[2].sigmoid()
            elif spec.get('classification_loss_type') == 'CrossEntropyLoss':
                sc = outs[0].softmax(-1)[..., :-1]
            else:
                sc = outs[0].sigmoid()
            model._classification_threshold = float(torch.quantile(sc.flatten()[:4_000_000].float(), 0.98))
            det = ev_time(lambda: model.detect(outs, meta), a.reps)
            counts = model.detect(outs, meta).counts.cpu()
        P = outs[0].shape[1]
        rec = dict(config=name, batch=a.n, input=[a.h, a.w], points_per_image=int(P),
                   forward_eager_ms=dict(p50=round(eager[0], 4), p95=round(eager[1], 4)),
                   forward_graph_ms=dict(p50=round(graph[0], 4), p95=round(graph[1], 4)),
                   detect_ms=dict(p50=round(det[0], 4), p95=round(det[1], 4)),
                   images_per_s_graph_plus_detect=round(a.n / ((graph[0] + det[0]) * 1e-3), 1),
                   candidates_per_image=int(counts[:, 0].float().mean()), kept_per_image=int(counts[:, 1].float().mean()))
        if not a.no_cpu:
            from test_sibling_oracle_golden import oracle_forward
            xc = x[:1].cpu()
            torch.set_num_threads(os.cpu_count())
            oracle_forward(name, xc)
            t0 = time.time()
            oracle_forward(name, xc)
            dt = time.time() - t0
            rec['cpu_oracle'] = dict(images_per_s=round(1.0 / dt, 3), cores=os.cpu_count(), kind='port',
                                     sample='1 image, 1 rep after warm-up, forward only')
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
