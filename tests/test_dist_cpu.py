"""Image-parallel sharding: world_size-2 gloo run of the batch partition + result
gather + global loss normaliser used by bench.py / lfd_amd.parallel, on CPU."""
import os
import subprocess
import sys

from conftest import ROOT

_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'lfd-a-light-and-fast-detector_amd'))
import torch, torch.distributed as dist
from lfd_amd import parallel
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
# 1. contiguous image shards cover the batch exactly once
lo, hi = parallel.shard_range(11, rank, world)
sizes = [None] * world
dist.all_gather_object(sizes, (lo, hi))
assert sizes == [(0, 6), (6, 11)], sizes
# 2. per-image result lists gathered in image order on every rank
local = [[[rank, 0.5, float(i), 0., 1., 1.]] for i in range(lo, hi)]
allr = parallel.gather_results(local, 11)
assert [r[0][2] for r in allr] == [float(i) for i in range(11)]
# 3. global loss normaliser: n_pos is summed over ranks before the division (reference computes the loss
#    once over the gathered outputs of all replicas: executor.py:198-200, lfd.py:340,383)
n_pos = torch.tensor([3.0 if rank == 0 else 5.0])
assert float(parallel.global_count(n_pos)) == 8.0
# 4. gradient averaging helper == mean over ranks
g = torch.full((4,), float(rank + 1))
parallel.allreduce_mean_([g])
assert torch.allclose(g, torch.full((4,), 1.5))
# 4b. the flat gradient buffer of lfd_amd.optim.SGD is the all-reduce bucket (one collective, mean over ranks)
from lfd_amd import optim
torch.manual_seed(0)
lin = torch.nn.Linear(5, 3)
opt = optim.SGD(lin.parameters(), lr=0.1, momentum=0.9)
opt.zero_grad()
(lin(torch.ones(2, 5)).sum() * float(rank + 1)).backward()
opt.allreduce_grads()
assert torch.allclose(lin.weight.grad, torch.full((3, 5), 2.0 * 1.5)) and torch.allclose(lin.bias.grad, torch.full((3,), 2.0 * 1.5))
assert lin.weight.grad.data_ptr() == opt._flat[0].g.data_ptr()
# 4c. lfd_amd.train.backward_and_update with a plain torch optimizer: gradients are averaged over the ranks before the step
from lfd_amd import train
w = torch.nn.Parameter(torch.zeros(3))
topt = torch.optim.SGD([w], lr=1.0)
train.backward_and_update(topt, (w * torch.tensor([1.0, 2.0, 3.0]) * float(rank + 1)).sum())
assert torch.allclose(w.detach(), -1.5 * torch.tensor([1.0, 2.0, 3.0])), w
# 4d. lfd_amd.train.SegmentedIteration -- the control flow of the graphed image-parallel training step (GraphedTrainStep under
#     torch.distributed: three captured segments, the iteration's two collectives between them) on a toy model whose
#     segments do what the real ones do: A local loss sums, B finalize with the GLOBAL normaliser + backward into the flat
#     gradient buffer, C divide by the world size + update.  Equals the one-process step over the concatenated batch.
torch.manual_seed(1)
W0 = torch.randn(3)
xs = [torch.tensor([[1., 2., 3.], [0., 1., 0.]]), torch.tensor([[2., 0., 1.]])]     # rank 0: 2 items, rank 1: 1 item
ys = [torch.tensor([1., 0.]), torch.tensor([2.])]
w = torch.nn.Parameter(W0.clone())
sopt = optim.SGD([w], lr=0.5)
S = {}
def seg_a():
    S['pred'] = xs[rank] @ w
    S['sums'].copy_(torch.tensor([float(xs[rank].size(0)), 0.0], dtype=torch.float64))      # local count (the "n_pos")
def seg_b():
    n_global = S['gsums'][0]                                   # reduced between A and B
    loss = ((S['pred'] - ys[rank]) ** 2).sum() / n_global * world     # rank's loss scaled by the world size (lfd.py get_loss)
    sopt.zero_grad()
    loss.backward()
def seg_c():
    sopt._flat[0].g /= world
    with torch.no_grad():
        w.sub_(0.5 * w.grad)                                   # (the update KERNEL is HIP-only; the flat buffer is the bucket)
S['sums'], S['gsums'] = torch.zeros(2, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)
calls = []
it = train.SegmentedIteration(lambda: (calls.append('a'), seg_a()), lambda: (calls.append('b'), seg_b()),
                              lambda: (calls.append('c'), seg_c()), S['sums'], S['gsums'], [sopt._flat[0].g])
it.run()
assert calls == ['a', 'b', 'c'] and float(S['gsums'][0]) == 3.0
wr = torch.nn.Parameter(W0.clone())
lr_ = ((torch.cat(xs) @ wr - torch.cat(ys)) ** 2).sum() / 3.0
lr_.backward()
assert torch.allclose(w.detach(), (wr - 0.5 * wr.grad).detach(), atol=1e-6), (w, wr)
it.run()                                                       # replays keep working on the same buffers
assert calls == ['a', 'b', 'c', 'a', 'b', 'c']
# 5. throughput aggregation: max time over ranks
t = parallel.max_over_ranks(1.0 + rank)
assert t == 2.0
dist.destroy_process_group()
print('ok', rank)
'''


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', '29577', str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count('ok') == 2
