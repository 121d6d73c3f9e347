"""One training iteration -- the loop body of Executor.train (lfd/execution/executor.py:191-211) with
OptimizerHook.after_train_iter (lfd/execution/hooks/optimizer_hook.py:26-36), image-parallel over ranks instead of
nn.DataParallel (executor.py:39): forward -> get_loss (global-batch normalisers) -> zero_grad -> backward ->
one all-reduce of the flat gradient buffer -> clip_grad_norm_ (first `duration` epochs) + SGD update.
"""
import numpy as np
import torch

from . import ops, optim, parallel, train_engine


class DynamicLossScale(object):
    """Loss scale of the fp16 activation-gradient path (train_engine / csrc/train.hip), driven by the finiteness of the
    total gradient norm -- the same signal the update kernel gates on (csrc/optim.hip: a non-finite norm skips the update):
    overflow -> halve (the skipped iteration costs one step, the weights stay clean); `growth_interval` clean iterations in a
    row -> double.  Powers of two in [min_scale, max_scale], so scaling and unscaling are exact.  The reference trains in fp32
    (optimizer_hook.py:26-36) and needs none; with fp32-representable gradients (|dz| < 65504 / scale) the results do not
    depend on the scale (measured 1 .. 65536)."""

    def __init__(self, init_scale=None, growth_interval=2000, min_scale=1.0, max_scale=65536.0):
        if init_scale is not None:
            train_engine.set_loss_scale(init_scale)
        self.growth_interval, self.min_scale, self.max_scale = int(growth_interval), float(min_scale), float(max_scale)
        self.good_steps, self.skipped = 0, 0

    @property
    def scale(self):
        return train_engine.loss_scale()

    def update(self, grad_norm):
        """grad_norm: the iteration's total gradient norm (float or 0-dim tensor).  -> True if the update was applied"""
        finite = bool(np.isfinite(float(grad_norm)))
        if not finite:
            self.skipped += 1
            self.good_steps = 0
            train_engine.set_loss_scale(max(self.min_scale, self.scale / 2))
            return False
        self.good_steps += 1
        if self.good_steps >= self.growth_interval:
            self.good_steps = 0
            train_engine.set_loss_scale(min(self.max_scale, self.scale * 2))
        return True


def train_step(model, optimizer, image_batch, annotation_batch, grad_clip_cfg=None, clip_active=True, loss_scaler=None):
    """-> (loss_values dict of floats, grad_norm).  grad_clip_cfg: dict(max_norm=..., norm_type=2) or None
    (config_dict['optimizer_grad_clip_cfg'] without 'duration'); clip_active: epoch < duration; loss_scaler: an optional
    DynamicLossScale (reads the norm: one more host sync)."""
    if loss_scaler is not None and not isinstance(optimizer, optim.SGD):
        # (a torch optimizer has no device-side overflow guard -- a skipped-step signal that is never raised would let
        #  the scale grow without bound and inf * 0 reach the weights)
        raise RuntimeError('train_step: a DynamicLossScale needs lfd_amd.optim.SGD (its update is skipped on the device when the '
                           'gradient norm is not finite)')
    predict_outputs = model(image_batch)
    loss_dict = model.get_loss(predict_outputs, annotation_batch)
    grad_norm = backward_and_update(optimizer, loss_dict['loss'], grad_clip_cfg, clip_active)
    if loss_scaler is not None:
        loss_scaler.update(optimizer.last_norm[0])
    return loss_dict['loss_values'], grad_norm


def backward_and_update(optimizer, loss, grad_clip_cfg=None, clip_active=True):
    optimizer.zero_grad()
    loss.backward()
    fused = isinstance(optimizer, optim.SGD)
    if fused:
        optimizer.allreduce_grads()
    elif parallel.is_dist():      # any other optimizer: average the gradients over the ranks as one flat bucket
        parallel.allreduce_mean_([p.grad for g in optimizer.param_groups for p in g['params'] if p.grad is not None])
    grad_norm = 0
    if grad_clip_cfg is not None and clip_active:
        if fused and float(grad_clip_cfg.get('norm_type', 2)) == 2.0:
            return optimizer.clip_and_step(grad_clip_cfg['max_norm'])
        import torch.nn.utils.clip_grad as clip_grad
        params = [p for g in optimizer.param_groups for p in g['params'] if p.requires_grad and p.grad is not None]
        if params:
            grad_norm = clip_grad.clip_grad_norm_(params, **grad_clip_cfg)
    optimizer.step()
    return grad_norm


class OptimizerHook(object):
    """Same constructor and after_train_iter(executor) contract as the reference hook (optimizer_hook.py:8-36):
    reads executor.config_dict['optimizer' | 'loss' | 'model' | 'epoch'], writes config_dict['grad_norm']."""

    def __init__(self, grad_clip_cfg, training_epochs):
        assert isinstance(grad_clip_cfg, dict) or grad_clip_cfg is None
        self._grad_clip_cfg = dict(grad_clip_cfg) if grad_clip_cfg is not None else None
        if self._grad_clip_cfg is not None:
            self._grad_clip_duration = self._grad_clip_cfg.pop('duration', training_epochs)
            assert self._grad_clip_duration > 0 and isinstance(self._grad_clip_duration, int)

    def after_train_iter(self, executor):
        cd = executor.config_dict
        active = self._grad_clip_cfg is not None and cd['epoch'] < self._grad_clip_duration
        cd['grad_norm'] = backward_and_update(cd['optimizer'], cd['loss'], self._grad_clip_cfg, active)


class PendingLossValues(object):
    """the three loss values of an iteration enqueued with GraphedTrainStep(..., sync=False); get() waits for that iteration.
    (Valid until three further iterations have been enqueued: the host slots are a ring.)"""

    def __init__(self, buf, done):
        self._buf, self._done = buf, done

    def get(self):
        self._done.synchronize()
        c, r, t = self._buf.tolist()
        return dict(loss=t, classification_loss=c, regression_loss=r)


class SegmentedIteration(object):
    """The image-parallel training iteration as THREE device segments with the iteration's two collectives between them
    (lfd/execution/executor.py:39,198-202: the reference's nn.DataParallel gathers the outputs, computes the loss once over
    the global batch -- lfd.py:340,383 `n_pos + 1` / `n_pos` -- and sums the replicas' gradients):

        A  forward -> targets -> this rank's loss sums           | all-reduce(sums)            (8 doubles)
        B  finalize (global normalisers) -> backward -> flat g   | all-reduce(flat gradients)  (one bucket per group)
        C  g /= world -> clip_grad_norm_ -> SGD update

    `a`, `b`, `c` are callables (eager segments, or `CUDAGraph.replay` of the captured ones: RCCL calls are not captured --
    the collectives run eagerly between the replays, on the stream the graphs replay on); `sums` / `gsums` the local and the
    reduced loss sums, `grad_buffers` the flat gradient buffers.  Device-agnostic: tests/test_dist_cpu.py drives it with
    gloo at world size 2."""

    def __init__(self, a, b, c, sums, gsums, grad_buffers):
        self.a, self.b, self.c = a, b, c
        self.sums, self.gsums, self.grad_buffers = sums, gsums, list(grad_buffers)

    def run(self):
        import torch.distributed as dist
        self.a()
        self.gsums.copy_(self.sums)
        if parallel.is_dist():
            dist.all_reduce(self.gsums, op=dist.ReduceOp.SUM)
        self.b()
        if parallel.is_dist():
            for g in self.grad_buffers:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
        self.c()


class GraphedTrainStep(object):
    """train_step as ONE HIP graph per set of optimizer hyper-parameters: forward, device target assignment, fused get_loss,
    the hand-written backward, clip_grad_norm_ + SGD -- ~500 launches replayed with one host call, then the iteration's one
    host sync (the three loss values).  Static shapes: every call must bring an image batch of the first call's shape and at
    most `max_boxes` annotations in total; the annotations are uploaded into fixed device buffers before the replay.

    Under torch.distributed (one process per GPU, image-parallel) the iteration is captured as THREE graphs split at its two
    collectives (SegmentedIteration: loss-normaliser sums, flat gradient bucket), replayed with the RCCL all-reduces between
    them -- the same kernels in the same order as the eager train_step of a rank, bit for bit (tests/test_gpu_dist.py).

    Covers what the all-HIP training path covers (train_engine.network_supported + the fused loss + lfd_amd.optim.SGD);
    raises otherwise -- use train_step.  Learning-rate schedules: the update kernel takes lr / momentum / weight decay
    by value, so a graph belongs to one set of values; a new set runs eagerly once and is captured when it repeats (a
    per-iteration warm-up schedule therefore stays eager, the constant-lr bulk of an epoch replays).

        step = GraphedTrainStep(model, optimizer, grad_clip_cfg=dict(max_norm=10, norm_type=2))
        loss_values, grad_norm = step(image_batch, annotation_batch)        # same contract as train_step
    """

    def __init__(self, model, optimizer, grad_clip_cfg=None, max_boxes=4096, max_graphs=4, loss_scaler=None):
        if not isinstance(optimizer, optim.SGD):
            raise RuntimeError('GraphedTrainStep needs lfd_amd.optim.SGD (the flat-buffer optimizer)')
        if grad_clip_cfg is not None and float(grad_clip_cfg.get('norm_type', 2)) != 2.0:
            raise RuntimeError('GraphedTrainStep: L2 gradient clipping only')
        self.model, self.opt = model, optimizer
        self.max_norm = None if grad_clip_cfg is None else float(grad_clip_cfg['max_norm'])
        self.max_boxes, self.max_graphs = int(max_boxes), int(max_graphs)
        self.graphs, self._last_key, self.x = {}, None, None
        self._eager_only = set()       # keys some rank could not capture (agreed by all-reduce): eager on every rank
        self.loss_scaler = loss_scaler      # DynamicLossScale or None; a graph belongs to one loss scale (it is in the key)

    # ------------------------------------------------------------------ static inputs
    def _bind(self, image_batch):
        m = self.model
        if not image_batch.is_cuda or not m.training:
            raise RuntimeError('GraphedTrainStep: a training-mode model and a device-resident image batch')
        if not train_engine.network_supported(m) or not m._fused_loss_supported(image_batch):
            raise RuntimeError('GraphedTrainStep: this model does not run on the all-HIP training path; use train_step')
        dev = image_batch.device
        self.x = torch.empty_like(image_batch).contiguous()
        # annotations: ONE device buffer [boxes f32 [max_boxes, 4] | labels i64 [max_boxes] | offsets i32 [n + 1]] filled by
        # ONE copy from a pinned host buffer of the same layout per iteration (three pageable copies cost ~0.2 ms of
        # host-side latency in front of every replay, tools/timing/train_trace.py)
        mb, n = self.max_boxes, image_batch.size(0)
        nbytes = 16 * mb + 8 * mb + 4 * (n + 1)
        self.ann = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.boxes = self.ann[:16 * mb].view(torch.float32).view(mb, 4)
        self.labels = self.ann[16 * mb:24 * mb].view(torch.int64)
        self.offs = self.ann[24 * mb:].view(torch.int32)
        # a ring of pinned staging buffers (a caller that does not read the loss of every iteration -- `sync=False` -- runs
        # ahead of the device: a staging buffer is rewritten only after the copy that read it has completed)
        self._stage = []
        for _ in range(3):
            hbuf = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
            hv = hbuf.numpy()
            self._stage.append(dict(buf=hbuf, boxes=hv[:16 * mb].view(np.float32).reshape(mb, 4), labels=hv[16 * mb:24 * mb].view(np.int64),
                                    offs=hv[24 * mb:].view(np.int32), done=None))
        self._stage_i = 0
        self._vals_host = [dict(buf=torch.zeros(3, dtype=torch.float32).pin_memory(), done=None) for _ in range(3)]
        self._vals_i = 0
        self.adesc = None

    def _upload(self, image_batch, annotation_batch):
        if self.x is None:
            self._bind(image_batch)
        if image_batch.shape != self.x.shape or image_batch.dtype != self.x.dtype:
            raise RuntimeError('GraphedTrainStep: image batch %s, captured for %s' % (tuple(image_batch.shape), tuple(self.x.shape)))
        if image_batch.data_ptr() != self.x.data_ptr():
            self.x.copy_(image_batch)             # (fill `step.x` in place to save this copy)
        if len(annotation_batch) != self.x.size(0):
            raise RuntimeError('GraphedTrainStep: one annotation per image')
        k = 0
        sg = self._stage[self._stage_i]
        self._stage_i = (self._stage_i + 1) % len(self._stage)
        if sg['done'] is not None:
            sg['done'].synchronize()
        h_boxes, h_labels, h_offs = sg['boxes'], sg['labels'], sg['offs']
        h_offs[0] = 0
        for i, (b, l) in enumerate(annotation_batch):
            b = np.asarray(b, dtype=np.float32).reshape(-1, 4)
            l = np.asarray(l, dtype=np.int64).reshape(-1)
            g = b.shape[0]
            if k + g > self.max_boxes:
                raise RuntimeError('GraphedTrainStep: more than max_boxes=%d annotations in the batch' % self.max_boxes)
            h_boxes[k:k + g] = b
            h_labels[k:k + g] = l
            k += g
            h_offs[i + 1] = k
        self.ann.copy_(sg['buf'], non_blocking=True)
        sg['done'] = torch.cuda.Event()
        sg['done'].record()

    # ------------------------------------------------------------------ one iteration, device side only
    def _iteration(self, clip):
        m = self.model
        cls, reg = m(self.x)
        if self.adesc is None:
            sizes = [m._head_indexes_to_feature_map_sizes[i] for i in range(m._num_heads)]
            self.adesc = ops.make_assign_desc(self.x.size(0), sizes, m._point_strides, m._regression_ranges, m._gray_ranges,
                                              m._num_classes, m._range_assign_mode, m._regression_loss_type == 'independent')
        d, total = self.adesc
        cls_t, reg_t = ops.assign_targets_device(d, total, m._num_classes, self.boxes, self.labels, self.offs)
        vals = m._fused_loss_tensor(cls, reg, cls_t, reg_t)
        self.opt.zero_grad()
        vals[2].backward()
        if clip:
            norm = self.opt.clip_and_step(self.max_norm)
        else:
            self.opt.step()
            norm = None
        return vals.detach(), norm, self.opt.last_norm

    # ------------------------------------------------------------------ the same iteration in three segments (image-parallel)
    def _segments(self, clip):
        """-> (SegmentedIteration of eager segments, state dict).  Segment B reads `state['gsums']` (reduced in place between
        A and B); the values the caller reads are state['vals'] / ['norm'] / ['nc'] after C."""
        m, opt = self.model, self.opt
        world = float(parallel.world_size())
        # d loss / d {classification_loss, regression_loss, loss}: the iteration differentiates `loss` (created here, outside
        # any capture: a host -> device copy is not capturable)
        S = {'g001': torch.tensor([0.0, 0.0, 1.0], dtype=torch.float32, device=self.x.device)}

        def seg_a():
            cls, reg = m(self.x)
            if self.adesc is None:
                sizes = [m._head_indexes_to_feature_map_sizes[i] for i in range(m._num_heads)]
                self.adesc = ops.make_assign_desc(self.x.size(0), sizes, m._point_strides, m._regression_ranges, m._gray_ranges,
                                                  m._num_classes, m._range_assign_mode, m._regression_loss_type == 'independent')
            d, total = self.adesc
            S['cls'], S['reg'] = cls, reg
            S['cls_t'], S['reg_t'] = ops.assign_targets_device(d, total, m._num_classes, self.boxes, self.labels, self.offs)
            S['desc'] = m._loss_desc(cls.size(0))
            sums = ops.get_loss_sums(S['desc'], cls.detach(), reg.detach(), S['cls_t'], S['reg_t'])
            if 'sums' not in S:
                S['sums'], S['gsums'] = sums, torch.empty_like(sums)
            elif S['sums'].data_ptr() != sums.data_ptr():
                S['sums'].copy_(sums)

        def seg_b():
            cls, reg = S['cls'], S['reg']
            fin = ops.get_loss_finalize(S['desc'], S['sums'], S['gsums'], world)
            gc, gr = ops.get_loss_backward(S['desc'], cls.detach(), reg.detach(), S['cls_t'], S['reg_t'], fin, S['g001'])
            opt.zero_grad()
            torch.autograd.backward([cls, reg], [gc.view_as(cls).to(cls.dtype), gr.view_as(reg).to(reg.dtype)])
            for fg in opt._flat:
                fg.adopt_grads()
            S['vals'] = fin[:3]

        def seg_c():
            if world != 1.0:
                for fg in opt._flat:
                    fg.g /= world
            if clip:
                S['norm'] = opt.clip_and_step(self.max_norm)
            else:
                opt.step()
                S['norm'] = None
            S['nc'] = opt.last_norm

        return seg_a, seg_b, seg_c, S

    def _run_dist(self, clip, key):
        """one image-parallel iteration: eager segments the first time a key shows up, then three captured graphs"""
        ent = self.graphs.get(key)
        if ent is None:
            a, b, c, S = self._segments(clip)
            import torch.distributed as dist
            # (every rank takes the same branch: the key holds the optimizer's hyper-parameters and the loss scale, identical on
            #  all ranks of an image-parallel run, and _eager_only is only ever extended by agreement, below)
            capture = self._last_key == key and key not in self._eager_only
            if capture:
                if len(self.graphs) >= self.max_graphs:
                    self.graphs.pop(next(iter(self.graphs)))
                torch.cuda.synchronize()
                stream = torch.cuda.Stream(device=self.x.device)
                ga, gb, gc_ = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                ok = [True]

                def captured(g, fn, pool):
                    """capture fn into g and run it once; a capture that fails on THIS rank runs the segment eagerly instead, so
                    that the rank still meets the others in the collective that follows -- the ranks agree on success below"""
                    if ok[0]:
                        try:
                            with torch.cuda.graph(g, pool=pool, stream=stream):
                                fn()
                            g.replay()
                            return
                        except Exception:
                            ok[0] = False
                    fn()
                # capture A, run the collective it feeds, capture B against the reduced sums, ... : every capture sees the
                # buffers in the state a replay finds them in; one memory pool, the replay order of the capture order
                captured(ga, a, None)
                it = SegmentedIteration(lambda: None, lambda: None, lambda: None, S['sums'], S['gsums'], [fg.g for fg in self.opt._flat])
                it.gsums.copy_(it.sums)
                dist.all_reduce(it.gsums, op=dist.ReduceOp.SUM)
                captured(gb, b, ga.pool() if ok[0] else None)
                for g in it.grad_buffers:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM)
                captured(gc_, c, ga.pool() if ok[0] else None)
                flag = torch.tensor([1 if ok[0] else 0], dtype=torch.int32, device=self.x.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                self._last_key = key
                self._bump_versions()
                if int(flag.item()) == 1:
                    it.a, it.b, it.c = ga.replay, gb.replay, gc_.replay
                    self.graphs[key] = (it, S, (ga, gb, gc_))
                else:
                    self._eager_only.add(key)      # some rank could not capture: every rank stays eager for this key
                return S['vals'], S['norm'], S['nc']
            it = SegmentedIteration(a, b, c, None, None, [fg.g for fg in self.opt._flat])
            # (eager: the sums tensor only exists after segment A)
            a()
            it.sums, it.gsums = S['sums'], S['gsums']
            it.a = lambda: None
            it.run()
            self._last_key = key
            return S['vals'], S['norm'], S['nc']
        it, S, _ = ent
        it.run()
        self._last_key = key
        self._bump_versions()
        return S['vals'], S['norm'], S['nc']

    def _bump_versions(self):
        # the replayed kernels rewrote parameters and norm buffers behind autograd's back: the inference engine keys its
        # packed-weight plans on the tensors' version counters
        optim.increment_version([p for grp in self.opt.param_groups for p in grp['params']])
        optim.increment_version(list(self.model.buffers()))

    def _key(self, clip):
        return (bool(clip), train_engine.loss_scale()) + tuple((float(g['lr']), float(g['momentum']), float(g['dampening']), float(g['weight_decay']),
                                      bool(g['nesterov'])) for g in self.opt.param_groups)

    def __call__(self, image_batch, annotation_batch, clip_active=True, sync=True):
        """sync=False: no host synchronisation -- returns (PendingLossValues, grad_norm device tensor or 0); `.get()` on the
        first waits for that iteration only.  A training loop that logs every k-th iteration keeps the device busy across
        iterations this way (the host-side gap between two synchronous replays is ~0.13 ms of a ~7 ms iteration)."""
        if not sync and self.loss_scaler is not None:
            raise RuntimeError('GraphedTrainStep: a DynamicLossScale reads the gradient norm on the host every iteration (sync=True)')
        self._upload(image_batch, annotation_batch)
        clip = self.max_norm is not None and clip_active
        key = self._key(clip)
        if parallel.is_dist():
            vals, norm, nc = self._run_dist(clip, key)
            return self._finish(vals, norm, nc, sync)
        ent = self.graphs.get(key)
        if ent is None and self._last_key == key:
            # second iteration in a row with these hyper-parameters (the first ran eagerly: momentum buffers, kernel
            # attributes, workspaces and weight-pack tables exist): capture, then replay like every later call
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                vals, norm, nc = self._iteration(clip)
            ent = (g, vals, norm, nc)
            self.graphs[key] = ent
        self._last_key = key
        if ent is None:
            vals, norm, nc = self._iteration(clip)
        else:
            ent[0].replay()
            vals, norm, nc = ent[1], ent[2], ent[3]
            self._bump_versions()
        return self._finish(vals, norm, nc, sync)

    def _finish(self, vals, norm, nc, sync):
        if not sync:
            slot = self._vals_host[self._vals_i]
            self._vals_i = (self._vals_i + 1) % len(self._vals_host)
            if slot['done'] is not None:
                slot['done'].synchronize()
            slot['buf'].copy_(vals, non_blocking=True)
            slot['done'] = torch.cuda.Event()
            slot['done'].record()
            return PendingLossValues(slot['buf'], slot['done']), (norm if norm is not None else 0)
        c, r, t = vals.tolist()          # the one host sync of the iteration
        if self.loss_scaler is not None:
            self.loss_scaler.update(nc[0])
        return dict(loss=t, classification_loss=c, regression_loss=r), (norm if norm is not None else 0)
