"""Image-parallel TRAINING STEP at world size 2 (gloo, CPU) against the reference's nn.DataParallel step on the same images
(lfd/execution/executor.py:39,198-202: the replicas run the forward on their image shards -- per-replica
BatchNorm batch statistics -- the outputs are gathered, `get_loss` runs ONCE over the whole batch (lfd.py:340,383: `n_pos + 1`,
`n_pos` are global-batch counts), and the replicas' gradients are summed into the one set of parameters).

What runs on the two ranks is the product's own step: LFD.forward (train mode, the CPU tensors' module path), LFD.get_loss
(lfd_amd/model/lfd.py:_loss_from_targets: all-reduced normalisers, the rank's loss scaled by the world size),
lfd_amd.train.backward_and_update with a torch optimizer (flat-bucket mean of the gradients, clip, step), and the flat gradient
buffer of lfd_amd.optim.SGD as the all-reduce bucket.  The two loss KERNELS have no CPU implementation in the product (like the
reference's CUDA-only extension): for this test their entry points are bound to differentiable fp32 restatements of
sigmoid_focal_loss_cuda.cu:24-97 and iou_loss.py:67-123 -- test infrastructure, the distributed logic around them is the
product's.  Comparator: ONE process that runs the two shards through the same modules (each shard its own forward, so each
its own batch statistics), concatenates the outputs, evaluates get_loss once, and lets autograd sum the gradients.

Also here: lfd_amd.parallel.sharded_map (what tools/infer_sharded.py runs) at world size 2 with an uneven split -- results in
image order, equal to the unsharded run."""
import os
import subprocess
import sys

from conftest import ROOT

_WORKER = r'''
import os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'lfd-a-light-and-fast-detector_amd')); sys.path.insert(0, os.path.join(root, 'tests', 'golden'))
import copy
import numpy as np
import torch, torch.distributed as dist
torch.set_num_threads(2)
from lfd_amd import configs, parallel, train, optim
import importlib
FL = importlib.import_module('lfd_amd.model.losses.focal_loss')
IL = importlib.import_module('lfd_amd.model.losses.iou_loss')
import train_step_cases as cases


# ---- CPU stand-ins for the two loss kernels (fp32, differentiable by autograd inside) ------------------------------------
def _focal_elems(x, t, gamma, alpha):
    # sigmoid_focal_loss_cuda.cu:24-59: label c positive for channel c, label == C (or any other) negative for every channel
    p = torch.sigmoid(x)
    c = torch.arange(x.size(1))[None]
    pos = (t[:, None] == c).float()
    neg = ((t[:, None] != c) & (t[:, None] >= 0)).float()
    t1 = -alpha * (1 - p).pow(gamma) * torch.log(p.clamp_min(1.1754943508222875e-38))
    t2 = -(1 - alpha) * p.pow(gamma) * (-x * (x >= 0).float() - torch.log1p(torch.exp(x - 2 * x * (x >= 0).float())))
    return pos * t1 + neg * t2


class _FocalExt(object):
    @staticmethod
    def forward(logits, targets, num_classes, gamma, alpha):
        return _focal_elems(logits.detach(), targets, gamma, alpha)

    @staticmethod
    def backward(logits, targets, d_losses, num_classes, gamma, alpha):
        x = logits.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            (_focal_elems(x, targets, gamma, alpha) * d_losses).sum().backward()
        return x.grad


def _iou_elems(pred, target, eps):
    # iou_loss.py:105-123 via bbox_overlaps(aligned) :286-321
    lt = torch.max(pred[:, :2], target[:, :2]); rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    a1 = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1]); a2 = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = torch.max(a1 + a2 - ov, torch.tensor(eps))
    return -torch.log((ov / union).clamp(min=eps))


def _iou_fwd(pred, target, eps):
    return _iou_elems(pred.detach(), target, eps)


def _iou_bwd(pred, target, d_loss, eps):
    x = pred.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        (_iou_elems(x, target, eps) * d_loss).sum().backward()
    return x.grad


FL._ext = _FocalExt
IL.ops.iou_loss_forward, IL.ops.iou_loss_backward = _iou_fwd, _iou_bwd

NAME, N, H, W = 'WIDERFACE_LFD_XS', 4, 96, 128
LR, MOM, WD, MAXN = 0.01, 0.9, 1e-4, 10.0


def model():
    m = configs.build_model(NAME)
    configs.perturb_weights(m, seed=1)
    return m.train()


x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(3)) * 2 - 1
rs = np.random.default_rng(9)
ann = []
for _ in range(N):
    g = int(rs.integers(1, 4))
    wh = np.exp(rs.uniform(np.log(8), np.log(70), (g, 2)))
    xy = rs.uniform(0, [W, H], (g, 2)) - wh / 2
    ann.append((np.concatenate([xy, wh], 1).astype(np.float32), np.zeros(g, np.int64)))

# ---- the reference's DataParallel step in ONE process (no process group yet: parallel.is_dist() is False) ------------------
ref = model()
ropt = torch.optim.SGD(ref.parameters(), lr=LR, momentum=MOM, weight_decay=WD)
ref_state0 = copy.deepcopy(ref.state_dict())
outs, stats_after_shard0 = [], None
for r in range(2):
    lo, hi = parallel.shard_range(N, r, 2)
    outs.append(ref(x[lo:hi]))                              # replica r: its own batch statistics
    if r == 0:
        stats_after_shard0 = {k: v.clone() for k, v in ref.state_dict().items() if 'running_' in k}
cls = torch.cat([o[0] for o in outs], 0); reg = torch.cat([o[1] for o in outs], 0)
rl = ref.get_loss((cls, reg), ann)                         # once over the gathered outputs
ropt.zero_grad(); rl['loss'].backward()
ref_grads = [p.grad.clone() for p in ref.parameters()]
ref_norm = float(torch.nn.utils.clip_grad_norm_(list(ref.parameters()), MAXN, 2))
ropt.step()
ref_params = [p.detach().clone() for p in ref.parameters()]

# ---- the same step, image-parallel over two ranks --------------------------------------------------------------------------
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 2 and parallel.is_dist()
m = model()
assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), ref_state0.values()))
opt = torch.optim.SGD(m.parameters(), lr=LR, momentum=MOM, weight_decay=WD)
lo, hi = parallel.shard_range(N, rank, world)
out = m(x[lo:hi])
ld = m.get_loss(out, ann[lo:hi])
# the rank's loss is world x (its sums / the GLOBAL normalisers): the mean over ranks is the reference's loss
t = torch.tensor([ld['loss_values'][k] for k in ('loss', 'classification_loss', 'regression_loss')], dtype=torch.float64)
dist.all_reduce(t); t /= world
want = torch.tensor([rl['loss_values'][k] for k in ('loss', 'classification_loss', 'regression_loss')], dtype=torch.float64)
assert torch.allclose(t, want, rtol=2e-6), (t, want)
# gradients before the update: flat-bucket mean over ranks == the reference's summed gradients
m.zero_grad(); ld['loss'].backward(retain_graph=True)
local = [p.grad.clone() for p in m.parameters()]
parallel.allreduce_mean_(local)
worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(local, ref_grads))
assert worst < 2e-5, worst
# the product's step: backward, flat-bucket mean, clip_grad_norm_, SGD
norm = float(train.backward_and_update(opt, ld['loss'], dict(max_norm=MAXN, norm_type=2), True))
assert abs(norm - ref_norm) <= 2e-5 * ref_norm, (norm, ref_norm)
worst_p = max(float((p.detach() - q).abs().max() / q.abs().max().clamp_min(1e-12)) for p, q in zip(m.parameters(), ref_params))
assert worst_p < 2e-6, worst_p
# both ranks hold the same parameters; BatchNorm running statistics stay per replica, rank 0's are the ones DataParallel
# keeps (replica 0 owns the module's buffers)
flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
both = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1])
if rank == 0:
    for k, v in m.state_dict().items():
        if 'running_' in k:
            assert torch.allclose(v, stats_after_shard0[k], rtol=1e-6, atol=1e-7), k
# the flat gradient buffer of lfd_amd.optim.SGD as the bucket, on the real model: same mean
m2 = model()
fopt = optim.SGD(m2.parameters(), lr=LR, momentum=MOM, weight_decay=WD)
ld2 = m2.get_loss(m2(x[lo:hi]), ann[lo:hi])
fopt.zero_grad(); ld2['loss'].backward(); fopt.allreduce_grads()
worst_f = max(float((p.grad - b).abs().max() / b.abs().max().clamp_min(1e-12)) for p, b in zip(m2.parameters(), ref_grads))
assert worst_f < 2e-5, worst_f
assert all(p.grad.data_ptr() == fopt._flat[0].g.data_ptr() + 4 * off for p, off in zip(fopt._flat[0].params, fopt._flat[0].offsets))

# ---- sharded inference launcher logic: uneven split, results in image order == the unsharded run ---------------------------
from oracle import net_oracle
arch = configs.ARCHS[NAME]
me = model().eval()
sd = {k: v.clone() for k, v in me.state_dict().items()}
frames = torch.rand(5, 3, 64, 96, generator=torch.Generator().manual_seed(4)) * 2 - 1


def detect_shard(lo, hi):
    res = []
    for i in range(lo, hi):
        with torch.no_grad():
            c, r, sizes = net_oracle.lfd_forward(sd, arch, frames[i:i + 1])
        thr = float(np.quantile(c.sigmoid().numpy(), 0.9))
        dets, labels, _, _ = net_oracle.get_results_single(c[0].numpy(), r[0].numpy(), sizes, net_oracle.strides_of(arch), arch, thr,
                                                           0.4, False, (64, 96), 1.0)
        res.append(net_oracle.pack_results(dets, labels))
    return res


sharded = parallel.sharded_map(5, detect_shard)
assert parallel.shard_range(5, rank, world) == ((0, 3) if rank == 0 else (3, 5))
whole = detect_shard(0, 5)
assert len(sharded) == 5 and sharded == whole and all(len(r) > 0 for r in whole)
dist.barrier()
dist.destroy_process_group()
print('ok', rank, 'grad', worst, 'param', worst_p, 'flat', worst_f)
'''


def test_world_size_2_training_step_equals_the_data_parallel_step(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29579', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29579', str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('ok') == 2
