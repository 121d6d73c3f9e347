"""Experiment: throughput of the whole-step HIP graph when TWO batches are in flight (two streams, separate activation /
output buffers) against the serial replay bench.py times.  Uses two model instances with identical weights so that nothing
in the engine has to change for the measurement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch  # noqa: E402
from lfd_amd import configs  # noqa: E402

B, H, W = 8, 1080, 1920
dev = torch.device('cuda', 0)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
models, xs, metas, streams = [], [], [], []
for i in range(depth):
    m = configs.build_model('WIDERFACE_LFD_S')
    configs.perturb_weights(m)
    m.eval().to(dev)
    m.use_graph = True
    m._classification_threshold = 0.745
    m._nms_cfg = dict(type='nms', iou_thr=0.4)
    models.append(m)
    xs.append((torch.rand(B, H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) * 2 - 1).half())
    metas.append(torch.tensor([[float(W), float(H), 1.0]] * B, dtype=torch.float32, device=dev))
    streams.append(torch.cuda.Stream(device=dev))
with torch.no_grad():
    for i in range(depth):
        with torch.cuda.stream(streams[i]):
            for _ in range(3):
                models[i].detect_resident(xs[i], metas[i])
    torch.cuda.synchronize()

    def run(n, k):
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for i in range(k):
                with torch.cuda.stream(streams[i]):
                    models[i].detect_resident(xs[i], metas[i])
            torch.cuda.synchronize()
        res = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(n):
                i = s % k
                with torch.cuda.stream(streams[i]):
                    models[i].detect_resident(xs[i], metas[i])
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / n * 1e3)
        res.sort()
        return res[len(res) // 2]

    serial = run(100, 1)
    piped = run(100, depth)
    a = models[0].detect_resident(xs[0], metas[0]); b = models[depth - 1].detect_resident(xs[depth - 1], metas[depth - 1])
    torch.cuda.synchronize()
    same = bool(torch.equal(a.counts, b.counts))
print(json.dumps(dict(depth=depth, serial_ms=round(serial, 4), pipelined_ms=round(piped, 4), speedup=round(serial / piped, 3),
                      images_per_s=round(B / piped * 1e3, 1), identical_counts=same)))
