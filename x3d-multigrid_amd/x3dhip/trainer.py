"""Training-step driver: flat parameter/gradient/momentum buffers, fused SGD, data-parallel
gradient all-reduce (RCCL via torch.distributed) and optional hipGraph capture of the step.

Replaces, for the hot path, what the reference does with ``nn.DataParallel`` + ``optim.SGD``
(train_x3d_kinetics_multigrid.py:177,183,244-279): one process per GPU, each rank runs the
same shape on its share of the batch, gradients are summed with one bucketed all-reduce of
the flat buffer (15.18 MB fp32 for X3D-M) and divided by the world size inside the fused SGD
kernel; BN statistics stay local to the rank (DataParallel semantics).
"""
import torch
import torch.nn.functional as F

from . import ops


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


class Trainer:
    def __init__(self, model, lr, momentum=0.9, weight_decay=5e-5, process_group=None, world_size=1,
                 use_graph=False):
        self.model = model
        self.fp = FlatParams(model)
        self.lr = lr
        self.momentum = momentum
        self.weight_decay = weight_decay
        self.pg = process_group
        self.world = world_size
        self.first = True
        self.use_graph = use_graph
        self._graphs = {}
        self._comm_stream = torch.cuda.Stream() if world_size > 1 else None

    # -- pieces ---------------------------------------------------------------------------
    def _fwd_bwd(self, x, y):
        self.fp.grad.zero_()
        logits = self.model(x)
        loss = F.cross_entropy(logits, y)
        loss.backward()
        return loss.detach(), logits.detach()

    def _allreduce(self):
        if self.world <= 1:
            return
        import torch.distributed as dist
        cur = torch.cuda.current_stream()
        self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            for (a, b) in self.fp.head_first_buckets(self.model):
                dist.all_reduce(self.fp.grad[a:b], op=dist.ReduceOp.SUM, group=self.pg)
        cur.wait_stream(self._comm_stream)

    def _sgd(self):
        ops.sgd_fused(self.fp.flat, self.fp.grad, self.fp.mom, self.lr, self.momentum, self.weight_decay,
                      1.0 / self.world, first=self.first)
        self.first = False

    # -- public -----------------------------------------------------------------------------
    def step(self, x, y):
        """One optimizer step on clips x[B,3,T,H,W], labels y[B,1].  Returns (loss, logits)."""
        if not self.use_graph:
            loss, logits = self._fwd_bwd(x, y)
            self._allreduce()
            self._sgd()
            return loss, logits
        return self._graphed_step(x, y)

    def _graphed_step(self, x, y):
        key = (tuple(x.shape), self.model.bn1.num_splits, self.model.training)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._capture(x, y)
            self._graphs[key] = ent
        ent["x"].copy_(x)
        ent["y"].copy_(y)
        ent["fb"].replay()
        self.model._pending_tracked += 1     # the replayed forward advanced every split-BN once
        self._allreduce()
        self._sgd()                          # outside the graph: lr / first-step flag are host values
        return ent["loss"], ent["logits"]

    def _capture(self, x, y):
        sx, sy = x.clone(), y.clone()
        # warm-up on a side stream (allocator + lazy init), restoring state that a real step mutates
        keep = self.fp.flat.clone()
        bn_state = {k: v.clone() for k, v in self.model.state_dict().items() if "running_" in k}
        pending = self.model._pending_tracked
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(sx, sy)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, logits = self._fwd_bwd(sx, sy)
        # undo the warm-up's side effects on BN running stats / counters
        self.fp.flat.copy_(keep)
        sd = self.model.state_dict()
        for k, v in bn_state.items():
            sd[k].copy_(v)
        self.model._pending_tracked = pending
        return dict(x=sx, y=sy, fb=g, loss=loss, logits=logits)
