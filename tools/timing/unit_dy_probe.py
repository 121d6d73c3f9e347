"""Per unit: the relative L2 error of the BatchNorm backward's dy against autograd GIVEN the stored tensors, split into the
elements whose ReLU decision agrees with PyTorch's and those where it does not (tests/test_gpu_train_convs.py (b))."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'lfd-a-light-and-fast-detector_amd'))
from lfd_amd import configs, train_engine

name, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nchw = lambda t: t.permute(0, 3, 1, 2).float()
torch.manual_seed(1)
m = configs.build_model(name).cuda().train()
configs.perturb_weights(m)
x = torch.randn(4, 3, h, w, device='cuda')
units, taps = train_engine.build_units(m._backbone)
tap_t, saved = train_engine.forward(units, taps, x)
acts, tape = saved
S = train_engine.LOSS_SCALE
ws = [torch.randn_like(nchw(t)) / t.numel() ** 0.5 for t in tap_t]
trace = []
train_engine.backward(units, saved, {t: (w_ * S).permute(0, 2, 3, 1).contiguous().half() for t, w_ in zip(taps, ws)}, trace=trace)
for rec in trace:
    u = units[rec['ui']]
    if not isinstance(u.norm, torch.nn.BatchNorm2d):
        continue
    y = nchw(tape[rec['ui']][0]).requires_grad_(True)
    gam, bet = u.norm.weight.detach().clone(), u.norm.bias.detach().clone()
    z = F.batch_norm(y, None, None, gam, bet, True, 0.1, u.norm.eps)
    if u.res is not None:
        z = z + nchw(acts[u.res])
    if u.relu:
        mask_t = z > 0
        z = F.relu(z)
    z.backward(nchw(rec['dz']) / S)
    dy = nchw(rec['dy']) / S
    err = dy - y.grad
    rel = float(err.norm() / y.grad.norm())
    line = 'unit %2d rel %.2e' % (rec['ui'], rel)
    if u.relu:
        mask_h = nchw(acts[u.dst]) > 0
        dis = mask_h != mask_t
        line += '  relu disagreements %d of %d; |dz| there max %.3e vs rms %.3e' % (
            int(dis.sum()), dis.numel(), float((nchw(rec['dz']) / S)[dis].abs().max()) if int(dis.sum()) else 0.0,
            float((nchw(rec['dz']) / S).pow(2).mean().sqrt()))
    print(line)
