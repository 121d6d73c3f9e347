"""tools/train_golden_diag.py -- the training iterations of the REAL reference (tests/golden/ref_train_step_*.npz) against
the two training routes of this package on the MI355X, iteration by iteration, written to gpurun_out/train_golden.json.

    python tools/train_golden_diag.py [case ...]

Routes: 'hip' = train_engine (fp16 NHWC activations, hand-written forward / backward, fused loss, flat SGD);
'torch' = LFD_HIP_TRAIN=0, the same mirror modules through PyTorch-ROCm fp32 autograd + torch.optim.SGD (what the CPU suite
pins to the reference bit for bit, here on MIOpen).  Per case and route: the three loss values and the gradient norm of every
iteration next to the reference's, the relative error of every BatchNorm running statistic after the last iteration (worst
five named), and for iteration 1 the outputs (max-abs / relative L2 vs the reference's cls / reg) and the per-parameter
gradient norms against the reference's summaries (worst five named).  Nothing here is a gate: tests/test_train_golden.py
holds the gates, this file is how their values were chosen."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)

from lfd_amd import configs, optim  # noqa: E402
import train_step_cases as cases  # noqa: E402


def summary(t):
    f = t.detach().double().reshape(-1).cpu()
    head = torch.zeros(4, dtype=torch.float64)
    head[:min(4, f.numel())] = f[:4]
    return np.concatenate([[float(f.norm()), float(f.mean())], head.numpy()])


def run(name, route, g):
    arch_name = cases.shape_of(name)[0]
    os.environ['LFD_HIP_TRAIN'] = '1' if route == 'hip' else '0'
    m = configs.build_model(arch_name)
    configs.perturb_weights(m, seed=1)
    m = m.train().cuda()
    if route == 'hip':
        opt = optim.SGD(m.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    else:
        opt = torch.optim.SGD(m.parameters(), lr=cases.LR, momentum=cases.MOMENTUM, weight_decay=cases.WEIGHT_DECAY)
    x = cases.images(name).cuda()
    ann = cases.annotations(name, configs.ARCHS[arch_name]['num_classes'])
    max_norm = float(cases.GRAD_CLIP['max_norm'])
    out = dict(losses=[], grad_norms=[], ref_losses=g['losses'].tolist(), ref_grad_norms=g['grad_norms'].tolist())
    names = [k for k, _ in m.named_parameters()]
    for it in range(cases.ITERATIONS):
        cls, reg = m(x)
        if it == 0:
            for key, t, stride in (('cls', cls, 1), ('reg', reg, 1), ('cls_s8', cls, 8), ('reg_s8', reg, 8)):
                if key in g.files:
                    a, b = t.detach().float().cpu().numpy()[:, ::stride], g[key]
                    out['fwd_' + key.split('_')[0]] = dict(max_abs=float(np.abs(a - b).max()),
                                                            rel_l2=float(np.linalg.norm(a - b) / np.linalg.norm(b)),
                                                            sig_max_abs=float(np.abs(1 / (1 + np.exp(-a)) - 1 / (1 + np.exp(-b))).max()))
        lo = m.get_loss((cls, reg), ann)
        opt.zero_grad()
        lo['loss'].backward()
        if it == 0:
            rows = []
            for (k, p), w in zip(m.named_parameters(), g['grad_summary']):
                gs = summary(p.grad)
                rows.append((k, float(gs[0]), float(w[0]), float(abs(gs[0] - w[0]) / max(w[0], 1e-12))))
            rows.sort(key=lambda r: -r[3])
            out['grad_norm_rel_err_worst'] = rows[:6]
            out['grad_norm_rel_err_median'] = float(np.median([r[3] for r in rows]))
            cos = []
            for k, p in m.named_parameters():
                if p.dim() <= 1 and ('grad/' + k) in g.files:
                    a, b = p.grad.detach().double().cpu().numpy().reshape(-1), g['grad/' + k].astype(np.float64).reshape(-1)
                    cos.append((k, float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30)), float(np.linalg.norm(b))))
            cos.sort(key=lambda r: r[1])
            out['grad_cos_1d_worst'] = cos[:6]
        if route == 'hip':
            gn = float(opt.clip_and_step(max_norm))
        else:
            gn = float(torch.nn.utils.clip_grad_norm_(list(m.parameters()), max_norm, 2))
            opt.step()
        lv = lo['loss_values']
        out['losses'].append([float(lv['loss']), float(lv['classification_loss']), float(lv['regression_loss'])])
        out['grad_norms'].append(gn)
        if it in (0, cases.ITERATIONS - 1):
            sd = m.state_dict()
            rows = []
            for k, v, w in zip(sd.keys(), sd.values(), g['state_summary_%d' % it]):
                if k.endswith('running_mean') or k.endswith('running_var'):
                    s = summary(v)
                    rows.append((k, float(s[0]), float(w[0]), float(abs(s[0] - w[0]) / max(w[0], 1e-3)), int(v.numel())))
            rows.sort(key=lambda r: -r[3])
            out['running_stats_rel_err_worst_it%d' % it] = rows[:6]
            out['running_stats_rel_err_median_it%d' % it] = float(np.median([r[3] for r in rows]))
    lr_, gr = np.array(out['losses']), np.array(out['ref_losses'])
    out['loss_rel_err'] = (np.abs(lr_ - gr) / np.maximum(np.abs(gr), 1e-12)).tolist()
    out['grad_norm_rel_err'] = (np.abs(np.array(out['grad_norms']) - g['grad_norms']) / g['grad_norms']).tolist()
    return out


def main():
    assert torch.cuda.is_available()
    names = sys.argv[1:] or (list(cases.CASES) + list(cases.LARGE_CASES))
    res = {}
    for name in names:
        f = os.path.join(ROOT, 'tests', 'golden', 'ref_train_step_%s.npz' % cases.file_tag(name))
        if not os.path.exists(f):
            continue
        g = np.load(f)
        res[name] = {}
        for route in ('hip', 'torch'):
            try:
                res[name][route] = run(name, route, g)
            except Exception as e:       # a route that does not cover a configuration is a finding, not a crash
                res[name][route] = dict(error=repr(e))
            r = res[name][route]
            print(name, route, 'loss rel err', np.round(r.get('loss_rel_err', []), 4).tolist(), 'grad-norm rel err',
                  np.round(r.get('grad_norm_rel_err', []), 4).tolist(), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'train_golden.json'), 'w') as fh:
        json.dump(res, fh, indent=1)


if __name__ == '__main__':
    main()
