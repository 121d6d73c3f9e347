"""Batch statistics the training forward stored (conv-epilogue sums) against fp64 sums over the stored y, unit by unit."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'lfd-a-light-and-fast-detector_amd'))
from lfd_amd import configs, train_engine

name, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(1)
m = configs.build_model(name).cuda().train()
configs.perturb_weights(m)
x = torch.randn(4, 3, h, w, device='cuda')
units, taps = train_engine.build_units(m._backbone)
_, (acts, tape) = train_engine.forward(units, taps, x)
for i, (u, (y, st)) in enumerate(zip(units, tape)):
    if not isinstance(u.norm, torch.nn.BatchNorm2d):
        continue
    c = y.size(3)
    yd = y.double().reshape(-1, c)
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    rstd = 1 / torch.sqrt(var + u.norm.eps)
    e_mean = float(((st[:c].double() - mean) * rstd).abs().max())       # in units of the channel's std
    e_rstd = float((st[c:].double() / rstd - 1).abs().max())
    print(i, tuple(y.shape), 'mean err / std %.2e   rstd rel err %.2e   max |mean|/std %.1f' % (e_mean, e_rstd, float((mean * rstd).abs().max())))
