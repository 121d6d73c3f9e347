"""tools/infer_sharded.py -- BASELINE.json configs[3]: TT100K_LFD_L, 45 classes, 32 x 1280x720 frames sharded
image-parallel over the GPUs of one node (4 frames per GPU on 8 GPUs), per-class NMS on the device, results gathered in
image order on every rank.  No data-path collective: each rank runs forward + decode + NMS on its own shard with
replicated weights; only the per-image result lists (a few KB) travel (lfd_amd.parallel.gather_results).

    python tools/infer_sharded.py                                   # one GPU, all 32 frames
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
        tools/infer_sharded.py                                      # RCCL (backend "nccl"), 4 frames per GPU

Replaces the reference's nn.DataParallel scatter / gather around LFD.forward (lfd/execution/executor.py:39,230-236).
Prints one JSON line on rank 0: frames, ranks, per-rank frames, detections, ms per sharded batch (max over ranks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='TT100K_LFD_L')
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--backend', default='nccl')
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or os.environ.get('LFD_FORCE_DIST') == '1':
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29611')
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from lfd_amd import configs, parallel
    m = configs.build_model(a.model)
    configs.perturb_weights(m)
    m.eval().to(dev)
    m.use_graph = True
    lo, hi = parallel.shard_range(a.frames, rank, world)
    # every rank draws the SAME 32 frames (seeded) and keeps its contiguous shard resident
    g = torch.Generator().manual_seed(0)
    frames = (torch.rand(a.frames, a.height, a.width, 3, generator=g) * 2 - 1).half()
    x = frames[lo:hi].contiguous().to(dev)
    meta = torch.tensor([[float(a.width), float(a.height), 1.0]] * (hi - lo), dtype=torch.float32, device=dev)
    with torch.no_grad():
        cls, _ = m.forward_resident(x)
        sc = cls[0].float().softmax(-1)[:, :-1] if m._is_ce() else cls[0].float().sigmoid()
        thr = float(torch.quantile(sc.reshape(-1)[::7], 1 - 2e-4))      # every rank: its own first frame -> all-reduce max
        thr = parallel.max_over_ranks(thr, dev)
        m._classification_threshold = thr
        m._nms_cfg = dict(type='nms', iou_thr=0.1)
        for _ in range(3):
            out = m.detect_resident(x, meta)
        torch.cuda.synchronize()
        if parallel.is_dist():
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = m.detect_resident(x, meta)
        torch.cuda.synchronize()
        dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)

        def detect_shard(lo_, hi_):       # this rank's frames are resident already: [lo, hi) is x
            assert (lo_, hi_) == (lo, hi)
            counts = out.counts.cpu()
            return [m._pack(out.dets[i, :int(counts[i, 1])], out.labels[i, :int(counts[i, 1])]) for i in range(hi - lo)]
    results = parallel.sharded_map(a.frames, detect_shard)      # every rank: all frames' rows, in frame order
    if rank == 0:
        print(json.dumps(dict(model=a.model, frames=a.frames, ranks=world, frames_per_rank=hi - lo, score_thr=thr,
                              detections=sum(len(r) for r in results), ms_per_batch=round(dt / a.steps * 1e3, 3),
                              images_per_s=round(a.frames * a.steps / dt, 1), backend=a.backend if parallel.is_dist() else None)))
    if parallel.is_dist():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
