"""SGD + gradient clipping over flat parameter buffers -- the device side of the reference's
OptimizerHook.after_train_iter (lfd/execution/hooks/optimizer_hook.py:26-36) with the optimizer the configs build
(torch.optim.SGD(params=model.parameters(), lr, momentum, weight_decay), WIDERFACE_LFD_S.py:216-226).

`SGD` keeps torch.optim.SGD's constructor, param_groups (the LR-scheduler hook writes group['lr']), state
('momentum_buffer' per parameter) and state_dict layout, but moves every parameter of a group into ONE contiguous
fp32 buffer (parameters become views of it; so do their .grad and momentum buffers).  A step is then

    [one RCCL all-reduce of the flat gradient buffer]  ->  norm (2 launches)  ->  clip + update (1 launch)

instead of ~3 ATen launches per tensor, and the flat gradient buffer IS the all-reduce bucket (no cat / copy-back).
The update itself has no CPU path: step / clip need float32 CUDA (ROCm) parameters and liblfd_hip.so.
"""
import ctypes as C

import torch
from torch.autograd.graph import increment_version

from . import parallel
from ._lib import check, lib, ptr, stream_ptr

__all__ = ['SGD', 'clip_grad_norm_']

_ALIGN = 64  # elements: every parameter starts on a 256-byte boundary


class _FlatGroup(object):
    """All parameters of one param group, contiguous."""

    def __init__(self, params):
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError('lfd_amd.optim.SGD needs float32 parameters on one device (got %s on %s)'
                                   % (p.dtype, p.device))
        self.params, self.offsets, o = params, [], 0
        for p in params:
            self.offsets.append(o)
            o += -(-p.numel() // _ALIGN) * _ALIGN
        self.numel = o
        self.device = dev
        self.p = torch.zeros(o, dtype=torch.float32, device=dev)
        self.g = torch.zeros(o, dtype=torch.float32, device=dev)
        self.m = torch.zeros(o, dtype=torch.float32, device=dev)
        self.norm_and_coef = torch.zeros(2, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.ws = None   # norm workspace, allocated on first use (needs liblfd_hip.so)
        self.rebound = False   # set when adopt_params() had to move / re-bind the buffers (SGD re-binds its state views)
        with torch.no_grad():
            for p, off in zip(params, self.offsets):
                v = self.view(self.p, p, off)
                v.copy_(p.data)
                p.data = v
                g = self.view(self.g, p, off)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                p._lfd_flat = (self, off)

    def require_device(self, what):
        """The buffers may be built on any device (the flat layout / all-reduce bucket is plain tensor plumbing,
        covered by the gloo tests), the kernels only run on the GPU."""
        if self.device.type != 'cuda':
            raise RuntimeError('lfd_amd.optim: %s runs as a HIP kernel and needs CUDA (ROCm) parameters; '
                               'there is no CPU path' % what)
        if self.ws is None:
            self.ws = torch.empty(lib().lfd_grad_norm_workspace_bytes(), dtype=torch.uint8, device=self.device)

    @staticmethod
    def view(flat, p, off):
        return flat[off:off + p.numel()].view(p.shape)

    def adopt_params(self):
        """Make sure every parameter is (still) the view of the flat parameter buffer.  `model.cuda()` / `.to()` /
        an assign-style load AFTER the optimizer was constructed (the reference builds the optimizer in
        prepare_optimizer() on CPU parameters and Executor moves the model afterwards, executor.py:36-39) replaces
        p.data behind the optimizer's back: the kernels would then update a buffer the model no longer reads.
        Re-flatten instead: the flat buffers follow the parameters to their device, the current values are copied
        in and the parameters are re-bound.  Returns True if anything had to be re-bound."""
        dev = self.params[0].device
        base = self.p.data_ptr()
        if dev == self.device and all(p.device == dev and p.data_ptr() == base + 4 * off and p.is_contiguous()
                                      for p, off in zip(self.params, self.offsets)):
            return False
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError('lfd_amd.optim.SGD needs float32 parameters on one device (got %s on %s)'
                                   % (p.dtype, p.device))
        with torch.no_grad():
            if dev != self.device:
                self.p = torch.zeros(self.numel, dtype=torch.float32, device=dev)
                self.g, self.m = self.g.to(dev), self.m.to(dev)
                self.norm_and_coef, self.sumsq = self.norm_and_coef.to(dev), self.sumsq.to(dev)
                self.ws = None
                self.device = dev
            for p, off in zip(self.params, self.offsets):
                v = self.view(self.p, p, off)
                if p.data_ptr() != v.data_ptr():
                    v.copy_(p.data)
                    p.data = v
        return True

    def adopt_grads(self):
        """Make sure every .grad is (still) the view of the flat gradient buffer; gradients that were replaced
        (zero_grad(set_to_none=True) followed by backward) are copied in.  Returns False if some are None."""
        if self.adopt_params():
            self.rebound = True
        complete = True
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is None:
                complete = False
                continue
            if g.device != self.device or g.data_ptr() != self.g.data_ptr() + 4 * off or not g.is_contiguous():
                v = self.view(self.g, p, off)
                v.copy_(g)
                p.grad = v
        return complete


def _flat_groups_of(parameters):
    """The flat groups covering exactly the given parameters, or None."""
    groups, seen = [], set()
    for p in parameters:
        fl = getattr(p, '_lfd_flat', None)
        if fl is None:
            return None
        if id(fl[0]) not in seen:
            seen.add(id(fl[0]))
            groups.append(fl[0])
    want = {id(p) for p in parameters}
    have = {id(p) for g in groups for p in g.params}
    return groups if want == have else None


def _norm(groups, max_norm):
    """Shared L2 norm of the flat gradient buffers -> the LAST group's norm_and_coef holds (norm, coef)."""
    extra = None
    for g in groups:
        g.require_device('clip_grad_norm_')
        with torch.cuda.device(g.device):
            check(lib().lfd_grad_norm_clip_coef_f32(ptr(g.g), g.numel, float(max_norm), ptr(extra), ptr(g.ws), g.ws.numel(),
                                                    ptr(g.norm_and_coef), ptr(g.sumsq), stream_ptr()),
                  'lfd_grad_norm_clip_coef_f32')
        extra = g.sumsq
    return groups[-1].norm_and_coef


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """torch.nn.utils.clip_grad_norm_ for parameters owned by an lfd_amd.optim.SGD (optimizer_hook.py:21-24):
    total L2 norm over the flat gradient buffers (fp64 accumulation), gradients scaled in place by
    min(max_norm / (norm + 1e-6), 1).  Returns the total norm as a 0-dim device tensor."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    if float(norm_type) != 2.0:
        raise NotImplementedError('lfd_amd.optim.clip_grad_norm_: only the L2 norm the configs use (norm_type=2)')
    groups = _flat_groups_of(parameters)
    if groups is None:
        raise RuntimeError('clip_grad_norm_: the parameters are not (all of) the parameters of an lfd_amd.optim.SGD; '
                           'construct the optimizer first')
    for g in groups:
        if not g.adopt_grads():
            raise RuntimeError('clip_grad_norm_: some parameters of the group have no gradient')
    nc = _norm(groups, max_norm)
    for g in groups:
        with torch.cuda.device(g.device):
            check(lib().lfd_scale_by_clip_coef_f32(ptr(g.g), g.numel, ptr(nc), stream_ptr()), 'lfd_scale_by_clip_coef_f32')
    if error_if_nonfinite and not bool(torch.isfinite(nc[0])):
        raise RuntimeError('The total norm for gradients is non-finite, so it cannot be clipped')
    return nc[0].clone()


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False):
        if lr < 0.0:
            raise ValueError('Invalid learning rate: {}'.format(lr))
        if momentum < 0.0:
            raise ValueError('Invalid momentum value: {}'.format(momentum))
        if weight_decay < 0.0:
            raise ValueError('Invalid weight_decay value: {}'.format(weight_decay))
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self._flat = [_FlatGroup([p for p in g['params'] if p.requires_grad]) for g in self.param_groups]

    # -- gradients ---------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """One memset per group; the .grad views stay in place (set_to_none is accepted and ignored: a None
        gradient would only make autograd allocate a fresh tensor that has to be copied back)."""
        for fg in self._flat:
            fg.g.zero_()
            fg.adopt_grads()
            for p, off in zip(fg.params, fg.offsets):
                if p.grad is None:
                    p.grad = fg.view(fg.g, p, off)

    def allreduce_grads(self):
        """Image-parallel training: average the gradients over ranks, one collective per group on the flat buffer
        (RCCL over xGMI; WF-S = 6.3 MB).  No-op without an initialised process group."""
        if not parallel.is_dist():
            return
        import torch.distributed as dist
        for fg in self._flat:
            fg.adopt_grads()
            dist.all_reduce(fg.g, op=dist.ReduceOp.SUM)
            fg.g /= dist.get_world_size()

    def _rebind_state(self, fg):
        """after _FlatGroup.adopt_params() moved the buffers: momentum buffers become views of the moved buffer again"""
        if not fg.rebound:
            return
        fg.rebound = False
        for p, off in zip(fg.params, fg.offsets):
            st = self.state.get(p)
            if st is not None and st.get('momentum_buffer') is not None:
                st['momentum_buffer'] = fg.view(fg.m, p, off)

    # -- update ------------------------------------------------------------------------------------------------
    def _step(self, nc, clip):
        """nc: device {total norm, clip coefficient} (the update is skipped on the device when the norm is not finite --
        the overflow guard of the fp16 gradient path, csrc/optim.hip); clip: multiply the gradients by the coefficient"""
        for group, fg in zip(self.param_groups, self._flat):
            ok = fg.adopt_grads()
            self._rebind_state(fg)
            if not ok:
                raise RuntimeError('lfd_amd.optim.SGD.step: a parameter has no gradient (call optimizer.zero_grad(), '
                                   'which keeps the flat gradient views, rather than setting .grad = None)')
            mom = float(group['momentum'])
            has = [('momentum_buffer' in self.state.get(p, {})) for p in fg.params]
            if mom != 0 and any(has) and not all(has):
                raise NotImplementedError('momentum buffers present for only some parameters of a group')
            first = mom != 0 and not any(has)
            fg.require_device('SGD.step')
            with torch.cuda.device(fg.device):
                check(lib().lfd_sgd_step_f32(ptr(fg.p), ptr(fg.g), ptr(fg.m), fg.numel, float(group['lr']), mom,
                                             float(group['dampening']), float(group['weight_decay']),
                                             int(bool(group['nesterov'])), int(first), ptr(nc), int(bool(clip)), 1,
                                             stream_ptr()),
                      'lfd_sgd_step_f32')
            if first:
                for p, off in zip(fg.params, fg.offsets):
                    self.state[p]['momentum_buffer'] = fg.view(fg.m, p, off)
            increment_version(fg.params)   # the kernel wrote the parameters behind autograd's back

    @torch.no_grad()
    def step(self, closure=None):
        """torch.optim.SGD.step -- with ONE difference that belongs to the fp16 activation-gradient path this optimizer was
        written for: the total gradient norm is computed every step (two small launches) and the update is SKIPPED on the
        device when it is not finite (an overflowed fp16 gradient would otherwise write inf / NaN into the weights); the norm
        and the clip coefficient stay in `last_norm` (device float32[2]: a non-finite last_norm[0] means "this step was
        skipped" -- lfd_amd.train.DynamicLossScale reads exactly that).  torch.optim.SGD has no such guard."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # the norm is computed even without clipping: it is the update's overflow guard (max_norm = inf -> coefficient 1)
        for fg in self._flat:
            fg.adopt_grads()
        self.last_norm = _norm(self._flat, float('inf'))
        self._step(self.last_norm, False)
        return loss

    @torch.no_grad()
    def clip_and_step(self, max_norm):
        """clip_grad_norm_(all parameters, max_norm) fused into the update: the clip coefficient stays on the device
        and is applied inside the update kernel (gradients are written back scaled, as clip_grad_norm_ leaves them).
        Returns the total norm (0-dim device tensor)."""
        for fg in self._flat:
            fg.adopt_grads()
        nc = _norm(self._flat, max_norm)
        self.last_norm = nc
        self._step(nc, True)
        return nc[0].clone()

    # -- checkpoints -------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for fg in self._flat:
                for p, off in zip(fg.params, fg.offsets):
                    buf = self.state[p].get('momentum_buffer') if p in self.state else None
                    if buf is not None:
                        v = fg.view(fg.m, p, off)
                        v.copy_(buf)
                        self.state[p]['momentum_buffer'] = v
