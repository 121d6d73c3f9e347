"""Training-step timing (BASELINE.json configs[4]: WIDERFACE_LFD_S, synthetic 640x640, bs 32 per GPU): forward, fused
get_loss, backward, gradient clipping + SGD.  Prints one JSON line per mode:
  hip   : the whole network on the hand-written kernels (train_engine) + fused loss + flat SGD
  graph : the same iteration replayed as one HIP graph (lfd_amd.train.GraphedTrainStep)
  torch : the same nn.Modules through PyTorch-ROCm autograd (LFD_HIP_TRAIN=0), op-by-op loss, torch.optim.SGD
Not the headline metric (bench.py is); recorded in DESIGN.md."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lfd-a-light-and-fast-detector_amd'))
from lfd_amd import configs, optim, train  # noqa: E402


def annotations(rng, n, hw, k=6):
    ann = []
    for _ in range(n):
        wh = np.exp(rng.uniform(np.log(8), np.log(200), (k, 2)))
        xy = rng.uniform(0, 1, (k, 2)) * (np.array([hw[1], hw[0]]) - wh).clip(1)
        ann.append((np.concatenate([xy, wh], 1).astype(np.float32), np.zeros(k, np.int64)))
    return ann


def run(mode, args):
    hip = mode in ('hip', 'graph')
    os.environ['LFD_HIP_TRAIN'] = '1' if hip else '0'
    os.environ['LFD_FUSED_LOSS'] = '1' if hip else '0'
    torch.manual_seed(0)
    m = configs.build_model(args.model).cuda().train()
    kw = dict(lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt = optim.SGD(m.parameters(), **kw) if hip else torch.optim.SGD(m.parameters(), **kw)
    rng = np.random.default_rng(0)
    x = torch.randn(args.batch, 3, args.size, args.size, device='cuda')
    ann = annotations(rng, args.batch, (args.size, args.size))
    clip = dict(max_norm=10, norm_type=2)
    if mode == 'graph':
        gstep = train.GraphedTrainStep(m, opt, clip, max_boxes=args.batch * 8)
        step = lambda: gstep(x if gstep.x is None else gstep.x, ann, True)      # noqa: E731
    else:
        step = lambda: train.train_step(m, opt, x, ann, clip, True)              # noqa: E731
    for _ in range(args.warmup):
        lv, _ = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lv, _ = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps(dict(mode=mode, model=args.model, batch=args.batch, size=args.size, ms_per_step=round(dt * 1e3, 3),
                          images_per_s=round(args.batch / dt, 1), loss=lv['loss'],
                          peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='WIDERFACE_LFD_S')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--modes', default='hip,torch')
    ap.add_argument('--concat', type=int, default=-1, help='train_engine.CONCAT_HEAD: 1 = shared head towers once over all pyramid '
                    'levels, 0 = level by level (default: the module\'s setting)')
    a = ap.parse_args()
    if a.concat >= 0:
        from lfd_amd import train_engine
        train_engine.CONCAT_HEAD = bool(a.concat)
    for md in a.modes.split(','):
        run(md, a)
