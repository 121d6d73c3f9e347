"""RCCL on the MI355X (backend "nccl" = RCCL on ROCm).  gpurun exposes ONE GPU, so what can be exercised here is a
world of size 1: process-group init on the device, the flat-gradient-bucket all-reduce of lfd_amd.optim.SGD, the
8-double loss-sum all-reduce of the fused get_loss, one full image-parallel training step and the sharded inference tool,
all through the real RCCL communicator.  The world_size-2 semantics (averaging, global normalisers) are covered by the
gloo test (tests/test_dist_cpu.py); 2/4/8-GPU runs are the driver's (bench.py --gpus N)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'lfd-a-light-and-fast-detector_amd'))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
from lfd_amd import configs, optim, parallel, train
assert parallel.is_dist() and parallel.world_size() == 1 and dist.get_backend() == 'nccl'
# 1. the collectives the training step issues, on device tensors through RCCL
t = torch.arange(8, dtype=torch.float64, device='cuda')
assert torch.equal(parallel.global_count(t), t)
m = configs.build_model('WIDERFACE_LFD_XS').cuda().train()
opt = optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
fg = opt._flat[0]
fg.g.copy_(torch.randn(fg.numel, device='cuda'))
before = fg.g.clone()
opt.allreduce_grads()                       # ONE all-reduce of the flat bucket (sum / world)
torch.cuda.synchronize()
assert torch.equal(fg.g, before)
# 2. a whole image-parallel training step with the process group live (global normalisers + gradient all-reduce)
x = torch.randn(2, 3, 96, 128, device='cuda')
ann = [(np.array([[10., 12., 30., 40.], [60., 20., 50., 44.]], np.float32), np.zeros(2, np.int64)),
       (np.array([[40., 30., 24., 20.]], np.float32), np.zeros(1, np.int64))]
lv, gn = train.train_step(m, opt, x, ann, dict(max_norm=10, norm_type=2), True)
assert np.isfinite(lv['loss']) and np.isfinite(float(gn))
dist.destroy_process_group()
# 3. the same step without a process group gives the same loss (world 1: normalisers and scale are identities)
torch.manual_seed(666)
m2 = configs.build_model('WIDERFACE_LFD_XS').cuda().train()
opt2 = optim.SGD(m2.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
lv2, _ = train.train_step(m2, opt2, x, ann, dict(max_norm=10, norm_type=2), True)
assert abs(lv2['loss'] - lv['loss']) <= 1e-5 * abs(lv['loss']), (lv, lv2)
print('rccl ok', lv['loss'])
'''


def test_rccl_world1_collectives_and_training_step(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'rccl ok' in out.stdout


def test_sharded_inference_tool_config4_one_rank_through_rccl():
    """tools/infer_sharded.py (BASELINE config 4 launcher) with the RCCL process group initialised at world size 1 and a
    reduced frame count: shard_range -> detect_resident -> gather_results."""
    env = dict(os.environ, LFD_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29633', HSA_ENABLE_IPC_MODE_LEGACY='0',
               RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'infer_sharded.py'), '--frames', '4', '--steps', '3'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    import json
    rec = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith('{')][-1])     # (RCCL prints its own lines)
    assert rec['frames'] == 4 and rec['ranks'] == 1 and rec['detections'] > 0 and rec['backend'] == 'nccl'
