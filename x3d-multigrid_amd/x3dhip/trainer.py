"""Training-step driver: flat parameter/gradient/momentum buffers, fused SGD, data-parallel
gradient all-reduce (RCCL via torch.distributed) and optional hipGraph capture of the step.

Replaces, for the hot path, what the reference does with ``nn.DataParallel`` + ``optim.SGD``
(train_x3d_kinetics_multigrid.py:177,183,244-279): one process per GPU, each rank runs the
same shape on its share of the batch, gradients are summed with one bucketed all-reduce of
the flat buffer (15.18 MB fp32 for X3D-M) and divided by the world size inside the fused SGD
kernel; BN statistics stay local to the rank (DataParallel semantics).
"""
import weakref

import torch
import torch.nn.functional as F

from . import ops

# every live Trainer of the process: captured graphs hold raw pointers into the process-wide finalize scratch (ops.scratch),
# so a retired scratch block may only be released when NO trainer holds a graph any more
_TRAINERS = weakref.WeakSet()


class FlatParams:
    """Re-homes every parameter of ``model`` into one contiguous fp32 buffer (and gives each a
    .grad view into a second one), in ``named_parameters()`` order."""

    def __init__(self, model):
        params = [p for _, p in model.named_parameters()]
        self.params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.mom = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = []
        o = 0
        for p in params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view(p.shape)
            p.grad = self.grad[o:o + k].view(p.shape)
            self.offsets.append((o, k))
            o += k
        self.numel = n

    def head_first_buckets(self, model, nbuckets=2):
        """Bucket boundaries (element ranges of the flat buffer) in reverse-autograd order:
        the head (fc1+fc2, 45 % of the bytes) is ready first, then layer4/3, then the rest."""
        names = [k for k, _ in model.named_parameters()]
        cut = None
        for i, k in enumerate(names):
            if k.startswith("layer3."):
                cut = self.offsets[i][0]
                break
        if cut is None or nbuckets < 2:
            return [(0, self.numel)]
        return [(cut, self.numel), (0, cut)]   # late layers + head first, early layers last


class GradReducer:
    """Data-parallel gradient exchange: sum the flat gradient over the ranks of ``pg`` with one
    all-reduce per bucket (RCCL on GPUs -- backend "nccl" is RCCL on ROCm; gloo in the CPU
    tests).  Buckets go out in reverse-autograd order on a side stream so that, when the
    backward pass is split around them, the early buckets travel over xGMI while the remaining
    backward kernels run.  The division by the world size happens in the fused SGD kernel."""

    def __init__(self, flat_grad, buckets, world_size, process_group=None, force_collectives=False):
        self.flat_grad, self.buckets, self.world, self.pg = flat_grad, buckets, world_size, process_group
        # force_collectives: issue the all-reduces on a one-rank group too (a sum over one rank is the identity) -- how
        # the RCCL path (backend "nccl") is exercised on a single-GPU box (tests/test_train_gpu.py)
        self.force_collectives = bool(force_collectives)
        self.active = world_size > 1 or (process_group is not None and self.force_collectives)
        self.stream = torch.cuda.Stream() if (self.active and flat_grad.is_cuda) else None

    def buffers_like(self):
        return list(self.buckets)

    def reduce(self):
        if not self.active:
            return
        import torch.distributed as dist
        if self.stream is None:                       # CPU tensors (gloo)
            for (a, b) in self.buckets:
                dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.pg)
            return
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for (a, b) in self.buckets:
                dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.pg)
        cur.wait_stream(self.stream)

    def start_bucket(self, i):
        """Enqueue the all-reduce of bucket i behind everything on the current stream WITHOUT making the current
        stream wait for it: the caller keeps launching backward kernels (x3dhip.trainer: the early layers) while
        the bucket travels over xGMI.  finish() joins."""
        if not self.active:
            return
        import torch.distributed as dist
        a, b = self.buckets[i]
        if self.stream is None:
            dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.pg)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self):
        if self.active and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class Trainer:
    """Optimizer-like object (``param_groups``, ``state_dict`` in torch.optim.SGD's format) that
    owns the whole training step."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=5e-5, process_group=None, world_size=1,
                 use_graph=False, num_steps_per_update=1, overlap=True, force_split=False, force_collectives=False):
        """overlap: multi-rank steps are captured as two graphs around the first gradient bucket (False: single graph,
        all-reduce after the whole backward); force_split: the two-graph form on a single rank too (tests);
        force_collectives: all-reduce on a one-rank process group (the single-GPU RCCL test).  Round 4: constructor
        arguments -- nothing in this module reads the environment."""
        self.model = model
        self.overlap, self.force_split = bool(overlap), bool(force_split)
        self.fp = FlatParams(model)
        # gradient accumulation (train_x3d_kinetics_multigrid.py:119,267-273): every train_step is one micro-batch with
        # loss / num_steps_per_update; the parameters move on every num_steps_per_update-th call
        self.num_steps_per_update = int(num_steps_per_update)
        self._micro = 0
        self.accum = torch.zeros_like(self.fp.grad) if self.num_steps_per_update > 1 else None
        self._acc_reducer = None
        self.stepped = False
        self.param_groups = [dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay,
                                  nesterov=False, params=list(range(len(self.fp.params))))]
        self.pg = process_group
        self.world = world_size
        self.first = True
        self.use_graph = use_graph
        self._graphs = {}
        _TRAINERS.add(self)
        model._direct_grads = True      # engine writes parameter gradients straight into fp.grad
        self.reducer = GradReducer(self.fp.grad, self.fp.head_first_buckets(model), world_size, process_group,
                                   force_collectives=force_collectives)

    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, v):
        self.param_groups[0]['lr'] = v

    # -- pieces ---------------------------------------------------------------------------
    def _head_loss_bwd(self, pooled, y):
        """fc1 / ReLU / dropout / fc2, mean cross entropy and their backward on the pooled rows [N, C5], straight into the
        flat gradient views (x3d.py:333-339, train...:259-271): HIP kernels only, no autograd."""
        model = self.model
        p = float(model.dropout.p)
        rng = model._head_rng(pooled.device) if p > 0 else None
        w1 = model.fc1.weight
        w1v = w1.data.view(w1.shape[0], -1)
        hd, logits = ops.head_fwd(pooled, w1v, model.fc2.weight.data, model.fc2.bias.data, p, rng)
        loss, dlog = ops.head_ce(logits, y.reshape(-1).contiguous(), rng)
        dpooled, _, _, _ = ops.head_bwd(dlog, hd, pooled, w1v, model.fc2.weight.data, p,
                                        outs=(w1.grad.view(w1.shape[0], -1), model.fc2.weight.grad, model.fc2.bias.grad))
        return loss.view(()), logits.unsqueeze(2), dpooled

    def _fwd_bwd(self, x, y):
        from . import engine
        model = self.model
        if getattr(model, "task", "class") != "class" or not model.training or not x.is_cuda:
            self.fp.grad.zero_()                # generic path through autograd (per-frame head, eval)
            logits = model(x)
            loss = F.cross_entropy(logits, y)
            loss.backward()
            return loss.detach(), logits.detach()
        self.fp.grad.zero_()
        tctx = engine.TrunkContext()
        with torch.no_grad():
            pooled = engine.trunk_forward(model, x.contiguous().float(), True, tctx)
            model._pending_tracked += 1
            loss, logits, dpooled = self._head_loss_bwd(pooled, y)
            side = engine.side_stream(x.device) if engine.use_side_stream() else None
            engine.trunk_backward(model, tctx, dpooled, engine._GradSink(True, side))
        return loss, logits

    def _allreduce(self):
        self.reducer.reduce()

    def _sgd(self, grad=None):
        g = self.param_groups[0]
        ops.sgd_fused(self.fp.flat, self.fp.grad if grad is None else grad, self.fp.mom, g['lr'], g['momentum'],
                      g['weight_decay'], 1.0 / self.world, first=self.first)
        self.first = False

    # -- public -----------------------------------------------------------------------------
    def step(self, x, y):
        """One optimizer step on clips x[B,3,T,H,W], labels y[B,1].  Returns (loss, logits)."""
        return self.train_step(x, y)

    def train_step(self, x, y, pre_step=None):
        """forward + CE + backward (+ all-reduce) + SGD.  ``pre_step`` runs after backward and
        before the parameter update (where the reference calls lr_warmup, train...:274).  With num_steps_per_update = K > 1
        every call is one micro-batch: its gradient / K is added to the accumulation buffer, and only the K-th call
        all-reduces, runs ``pre_step`` and updates the parameters (``self.stepped`` tells which kind of call it was)."""
        if self.num_steps_per_update > 1:
            return self._train_step_accum(x, y, pre_step)
        self.stepped = True
        if self.use_graph and self._overlap():
            loss, logits = self._graphed_split(x, y)        # all-reduce issued inside, overlapped with the early layers
        else:
            if self.use_graph:
                loss, logits = self._graphed_fwd_bwd(x, y)
            else:
                loss, logits = self._fwd_bwd(x, y)
            self._allreduce()
        if pre_step is not None:
            pre_step()
        self._sgd()                          # outside any graph: lr / first-step flag are host values
        return loss, logits

    @staticmethod
    def _feed(ent, x, y):
        """Hand a batch to a captured graph: copied into the graph's static input tensors -- unless the caller already wrote
        it there (`static_inputs`): a device-side input pipeline (kinetics_multigrid.DeviceVideoKinetics(out=...)) or a
        benchmark that keeps its synthetic clips resident fills the static buffers directly and no copy is made."""
        if x.data_ptr() != ent["x"].data_ptr():
            ent["x"].copy_(x)
        if y.data_ptr() != ent["y"].data_ptr():
            ent["y"].copy_(y)

    def static_inputs(self, x_shape):
        """(clips, labels) tensors the captured graph(s) of this clip shape read, or None before the first step of that
        shape: write the next batch into them and pass them to train_step to skip the input copy."""
        shp = tuple(x_shape)
        for key, ent in self._graphs.items():
            if (key[0] == "split" and key[1] == shp) or key[0] == shp:
                return ent["x"], ent["y"]
        return None

    MAX_GRAPHS = 6      # shapes of one long cycle: 2-3 (cycle_batch_sampler.py:98-111); each graph pins its activations

    def _lookup(self, key):
        """Graph cache: entries captured under an older split-BN layout point at split_bn buffers that
        update_bn_splits_long_cycle has since re-created (x3d.py:298-303) -- they are dropped as soon as the version moves,
        so a long-cycle switch releases their private pools (all activations of a step) instead of stranding them; the
        cache is also bounded (least recently used first)."""
        ver = self.model._bn_version
        if getattr(self, "_graphs_version", ver) != ver and self._graphs:
            self.invalidate_graphs()
        self._graphs_version = ver
        ent = self._graphs.pop(key, None)
        if ent is not None:
            self._graphs[key] = ent             # most recently used last
        return ent

    def _store(self, key, ent):
        while len(self._graphs) >= self.MAX_GRAPHS:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = ent

    def _train_step_accum(self, x, y, pre_step):
        K = self.num_steps_per_update
        loss, logits = self._graphed_fwd_bwd(x, y) if self.use_graph else self._fwd_bwd(x, y)
        ops.grad_accumulate(self.accum, self.fp.grad, 1.0 / K, first=self._micro == 0)
        self._micro += 1
        self.stepped = self._micro == K
        if self.stepped:
            self._micro = 0
            if self.reducer.active:
                # one exchange per optimizer step, on the accumulated gradient (no overlap with a backward pass here)
                if self._acc_reducer is None:
                    self._acc_reducer = GradReducer(self.accum, self.reducer.buffers_like(), self.world, self.pg,
                                                    force_collectives=self.reducer.force_collectives)
                self._acc_reducer.reduce()
            if pre_step is not None:
                pre_step()
            self._sgd(self.accum)
        return loss, logits

    def _graphed_fwd_bwd(self, x, y):
        key = (tuple(x.shape), self.model._bn_version, self.model.training)
        ent = self._lookup(key)
        if ent is None:
            ent = self._capture(x, y)
            self._store(key, ent)
        self._feed(ent, x, y)
        ent["fb"].replay()
        self.model._pending_tracked += 1     # the replayed forward advanced every split-BN once
        return ent["loss"], ent["logits"]

    # -- data parallel: backward captured as two graphs around the first gradient bucket ---------
    def _overlap(self):
        """Split capture is used for multi-rank runs (Trainer(overlap=False) disables it; force_split=True enables it on a
        single rank for tests), task 'class', with the two-bucket layout."""
        if not self.overlap or len(self.reducer.buckets) != 2:
            return False
        if getattr(self.model, "task", "class") != "class" or not self.model.training:
            return False
        return self.world > 1 or self.force_split

    def _fwd_bwd_late(self, x, y):
        """Graph A: zero grads, forward, loss, head backward, trunk backward of conv5 / layer4 / layer3."""
        from . import engine
        model = self.model
        self.fp.grad.zero_()
        tctx = engine.TrunkContext()
        with torch.no_grad():
            pooled = engine.trunk_forward(model, x.contiguous().float(), True, tctx)
            loss, logits, dpooled = self._head_loss_bwd(pooled, y)                               # x3d.py:333-339
            side = engine.side_stream(x.device) if engine.use_side_stream() else None
            sink = engine._GradSink(True, side)
            state = engine.trunk_backward(model, tctx, dpooled, sink, part="late")
        return loss, logits, tctx, sink, state

    def _capture_split(self, x, y):
        from . import engine
        sx, sy = x.clone(), y.clone()
        bn_state = {k: v.clone() for k, v in self.model.state_dict().items() if "running_" in k}
        pending = self.model._pending_tracked
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):                                  # allocator / lazy-init warm-up, eager
                _, _, tctx, sink, state = self._fwd_bwd_late(sx, sy)
                engine.trunk_backward(self.model, tctx, None, sink, part="early", state=state)
        torch.cuda.current_stream().wait_stream(s)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread-local capture: the process group's watchdog thread may touch the HIP runtime while we capture
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            loss, logits, tctx, sink, state = self._fwd_bwd_late(sx, sy)
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            engine.trunk_backward(self.model, tctx, None, sink, part="early", state=state)
        self.model._pending_tracked = pending
        sd = self.model.state_dict()
        for k, v in bn_state.items():
            sd[k].copy_(v)
        return dict(x=sx, y=sy, ga=ga, gb=gb, loss=loss, logits=logits, keep=(tctx, sink, state))

    def _graphed_split(self, x, y):
        key = ("split", tuple(x.shape), self.model._bn_version)
        ent = self._lookup(key)
        if ent is None:
            ent = self._capture_split(x, y)
            self._store(key, ent)
        self._feed(ent, x, y)
        ent["ga"].replay()
        self.reducer.start_bucket(0)          # head + layer4 + layer3 gradients: on the wire while graph B runs
        ent["gb"].replay()
        self.reducer.start_bucket(1)
        self.reducer.finish()
        self.model._pending_tracked += 1
        return ent["loss"], ent["logits"]

    def invalidate_graphs(self):
        """Captured graphs hold pointers to split_bn buffers: drop them when those are re-created
        (update_bn_splits_long_cycle) or when parameters are re-homed.  Called from the long-cycle switch
        (train_x3d_kinetics_multigrid.run) and, as a safety net, by the cache lookup when the BN version moved."""
        self._graphs.clear()
        if not any(t._graphs for t in _TRAINERS):
            # retired finalize-scratch blocks are shared by every graph of the process: freed only when none is left
            ops.release_retired_scratch()
        if torch.cuda.is_available():
            # hand the dropped graphs' private pools (all activations of a step per captured shape) back to the device.  This
            # synchronises the device -- at a long-cycle switch, i.e. 13 times over the whole schedule (every rank switches
            # at the same step: the sampler is deterministic in the step counter, cycle_batch_sampler.py:76-93)
            torch.cuda.empty_cache()

    def _capture(self, x, y):
        sx, sy = x.clone(), y.clone()
        # warm-up on a side stream (allocator + lazy init), restoring state that a real step mutates
        bn_state = {k: v.clone() for k, v in self.model.state_dict().items() if "running_" in k}
        pending = self.model._pending_tracked      # (state_dict() above flushed it: 0)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(sx, sy)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local" if self.world > 1 else "global"):
            loss, logits = self._fwd_bwd(sx, sy)
        # undo the warm-up's side effects on BN running stats / counters
        self.model._pending_tracked = pending
        sd = self.model.state_dict()
        for k, v in bn_state.items():
            sd[k].copy_(v)
        return dict(x=sx, y=sy, fb=g, loss=loss, logits=logits)

    # -- torch.optim.SGD-compatible state (reference checkpoints, train...:185-187,286-291) ---
    def state_dict(self):
        state = {}
        if not self.first:
            for i, (o, k) in enumerate(self.fp.offsets):
                state[i] = {'momentum_buffer': self.fp.mom[o:o + k].view(self.fp.params[i].shape).clone()}
        return {'state': state, 'param_groups': [dict(self.param_groups[0])]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        for k in ('lr', 'momentum', 'weight_decay'):
            if k in g:
                self.param_groups[0][k] = g[k]
        st = sd.get('state', {})
        if st:
            for i, (o, k) in enumerate(self.fp.offsets):
                buf = st.get(i, st.get(str(i)))
                if buf is not None and buf.get('momentum_buffer') is not None:
                    self.fp.mom[o:o + k].copy_(buf['momentum_buffer'].reshape(-1).to(self.fp.mom.device))
            self.first = False
