"""Headline benchmark: X3D-M training step (forward + loss + backward + fused SGD) at the
multigrid base shape B=8, T=16, H=W=224 (BASELINE.json configs[1]) on N MI355X GPUs, one
process per GPU, weak scaling (per-GPU batch fixed), synthetic NCTHW clips resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0): clips/s over the whole job, plus
  roofline     -- the dominant kernel family (by device time in one instrumented step):
                  algorithmic bytes (in+out elements of each conv pass) over
                  its HIP-event-measured duration, vs the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (stock-PyTorch restatement of the reference) timed on the
                  host cores on the stated workload (B=8 steps, bounded to ~1 minute), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "x3d-multigrid_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


_ARCH = {  # (cm, co, blocks) per stage: x3d.py:352-363 ("L": XL depth at M width)
    "S": [(54, 24, 3), (108, 48, 5), (216, 96, 11), (432, 192, 7)],
    "M": [(54, 24, 3), (108, 48, 5), (216, 96, 11), (432, 192, 7)],
    "L": [(54, 24, 5), (108, 48, 10), (216, 96, 25), (432, 192, 15)],
    "XL": [(72, 32, 5), (162, 72, 10), (306, 136, 25), (630, 280, 15)],
}


def algorithmic_elems_M(T, H, version="M", split=False):
    """E(T,H,W) of SURVEY.md 8(d) (elements per clip): sum over every Conv3d of in+out elements + 3 x block outputs.
    split=True: (E, E_wide) with E_wide the part of E that touches the wide (planes-channel) tensors inside the
    bottlenecks -- the ones the mixed-storage mode keeps in bf16."""
    def o(h):
        return (h - 1) // 2 + 1
    h = [H]
    for _ in range(5):
        h.append(o(h[-1]))
    S = [T * v * v for v in h]          # S[0]=input res, S[1]=stem res, S[2..5]=stage outputs
    arch = _ARCH[version]
    c0 = arch[0][1]
    E = 3 * S[0] + c0 * S[1] + 2 * c0 * S[1]       # conv1_s in+out, conv1_t in+out
    Ew = 0
    cin = c0
    for k, (cm, co, n) in enumerate(arch):
        sp, sk = S[k + 1], S[k + 2]
        # first block: conv1 (cin->cm @sp), conv2 (cm @sp -> @sk), conv3 (cm->co @sk), downsample (cin @sk sampled -> co @sk)
        E += (cin + cm) * sp + cm * (sp + sk) + (cm + co) * sk + cin * sp + co * sk
        E += (n - 1) * ((co + cm) * sk + 2 * cm * sk + (cm + co) * sk)
        E += 3 * n * co * sk
        Ew += cm * sp + cm * (sp + sk) + cm * sk + (n - 1) * 4 * cm * sk
        cin = co
    E += (arch[3][1] + arch[3][0]) * S[5] + arch[3][0] + 2048
    return (E, Ew) if split else E


class KernelTimer:
    """Brackets every ops.* launch of one eager step with HIP events on the current stream."""

    def __init__(self, ops):
        self.ops = ops
        self.records = []
        self._orig = {}

    # ops that are a fused form of another family's pass: reported under that family
    ALIAS = {"pw_bwd_data_res": "pw_bwd_data", "dw333_fwd_stats": "dw333_fwd"}

    def __enter__(self):
        self.pending_wbytes, self.pending_wjobs = 0, 0
        flush0 = self.ops.DeferredGrads.flush
        timer = self

        def flush(d):
            # the postponed weight gradients: every conv's kernel + the group sums, a few launches for the whole pass
            if not d.wjobs and not d.reduces:
                return flush0(d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flush0(d)
            e1.record()
            timer.records.append(("pw_bwd_weight", e0, e1, timer.pending_wbytes, "batched x%d" % timer.pending_wjobs,
                                  "pw_wgrad_batch(+reduce)"))
            timer.pending_wbytes, timer.pending_wjobs = 0, 0

        self._flush0 = flush0
        self.ops.DeferredGrads.flush = flush
        for name in ("pw_fwd", "pw_bwd_data", "pw_bwd_data_res", "pw_bwd_weight", "pw_bwd_fused", "dw333_fwd", "dw333_fwd_stats", "dw333_bwd", "stem133_fwd",
                     "stem133_bwd_weight", "dw5t_fwd", "dw5t_bwd", "bn_add_relu_fwd", "bn_stats_add_relu_fwd", "bn_add_relu_bwd",
                     "bn_relu_pool_fwd", "bn_relu_pool_bwd", "bn_fwd_finalize", "bn_bwd_finalize", "se_fwd",
                     "se_bn_bwd_finalize", "sgd_fused"):
            fn = getattr(self.ops, name)
            self._orig[name] = fn
            setattr(self.ops, name, self._wrap(name, fn))
        return self

    # entry points whose kernel template varies with the shape: the library reports which one it launched (x3d_last_kernel)
    KERNEL_OF = {"pw_fwd", "pw_bwd_data", "pw_bwd_data_res", "pw_bwd_fused", "dw333_fwd", "dw333_fwd_stats", "dw333_bwd"}

    def _wrap(self, name, fn):
        from x3dhip import _lib

        def inner(*a, **k):
            if name == "pw_bwd_weight" and k.get("defer") is not None:       # runs inside DeferredGrads.flush
                self.pending_wbytes += _alg_bytes(name, a, k, None)
                self.pending_wjobs += 1
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            shp = "x".join(map(str, a[0].shape)) if hasattr(a[0], "shape") else ""
            if name == "pw_bwd_fused":
                shp += " w=" + "x".join(map(str, a[3][:2]))
            elif name.startswith("pw_"):
                shp += " w=" + "x".join(map(str, (a[1] if name == "pw_fwd" else a[3]).shape[:2])) if name != "pw_bwd_weight" else " w=" + "x".join(map(str, a[4][:2]))
            kern = _lib.last_kernel() if name in self.KERNEL_OF else name
            self.records.append((self.ALIAS.get(name, name), e0, e1, _alg_bytes(name, a, k, r), shp, kern))
            return r
        return inner

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self.ops, name, fn)
        self.ops.DeferredGrads.flush = self._flush0

    def summary(self):
        torch.cuda.synchronize()
        agg, self.by_kernel = {}, {}
        self.launches = []
        for name, e0, e1, nbytes, shp, kern in self.records:
            ms_ = e0.elapsed_time(e1)
            self.launches.append((name, shp, round(ms_, 4), round(nbytes / (ms_ * 1e-3) / 1e9, 1) if ms_ > 0 else 0, kern))
            for key, table in ((name, agg), (kern, self.by_kernel)):
                d = table.setdefault(key, [0.0, 0, 0])
                d[0] += ms_
                d[1] += nbytes
                d[2] += 1
        return agg


def _alg_bytes(name, a, k, r):
    """Algorithmic bytes of one launch: 4 B x (input elements + output elements) of the conv /
    elementwise pass it implements (what the pass must move at minimum)."""
    n = lambda t: t.numel() * t.element_size()          # bytes as stored (4 per element; 2 for the bf16 wide tensors)
    try:
        if name in ("pw_fwd", "dw333_fwd", "dw333_fwd_stats", "dw5t_fwd", "stem133_fwd"):
            x, y = a[0], (r[0] if isinstance(r, tuple) else r)
            stride = k.get("stride", 1)
            xin = n(x) // (stride * stride) if name == "pw_fwd" else n(x)
            return xin + n(y)
        if name in ("pw_bwd_data", "pw_bwd_data_res"):
            return n(a[0]) + n(r[0])
        if name == "pw_bwd_fused":                      # data-gradient pass + weight-gradient pass of SURVEY 8(d), one launch
            return (n(a[0]) + n(r[0])) + (n(a[0]) + n(a[5]))
        if name == "pw_bwd_weight":
            stride = k.get("stride", 1)
            return n(a[0]) + n(a[3]) // (stride * stride)
        if name == "dw333_bwd":
            return 2 * (n(a[0]) + n(a[4]))              # fused data + weight pass
        if name == "dw5t_bwd":
            return 2 * (n(a[0]) + n(a[4]))
        if name == "stem133_bwd_weight":
            return n(a[0]) + n(a[1])
        if name in ("bn_add_relu_fwd", "bn_stats_add_relu_fwd", "bn_add_relu_bwd"):
            return 3 * n(a[0])
        if name in ("bn_relu_pool_fwd", "bn_relu_pool_bwd"):
            return n(a[0])
        if name == "sgd_fused":
            return 5 * n(a[0])
    except Exception:
        pass
    return 0


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_share():
    """Cores this process may really use: the cgroup CPU quota when there is one (the GPU box gives one GPU a 16-core share
    of a 256-thread host, and neither os.cpu_count() nor the affinity mask shows it), else the affinity mask; capped at 16,
    the share the pool documents for one GPU."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(T, H, B=8, sample_B=2):
    """CPU oracle (stock-PyTorch restatement of the reference, oracle/x3d_oracle.py) on the host cores: fwd+bwd of X3D-M ON THE
    STATED WORKLOAD (B = 8 clips of the headline shape; round 4 -- earlier rounds timed a B = 2 sample only), at ALL cores of
    this process's CPU share (_cpu_share) and at 8 threads (the 8-thread figure ties back to BASELINE.md's
    measurement of the reference itself in the 8-core build container).  Bounded: one warm-up + 3 steps at all cores, one
    warm-up + 2 steps at 8 threads, plus the old B = 2 sample (3 steps) for continuity with BENCH_r01..r03 -- about a minute."""
    from oracle import x3d_oracle as xo
    from x3dhip import synthetic
    avail = _cpu_share()
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, 1), 0)

    def timed(threads, batch, steps):
        x = synthetic.synthetic_clips(batch, T, H, H)
        y = synthetic.synthetic_labels(batch)
        torch.set_num_threads(threads)
        print("[bench] cpu baseline: oracle fwd+bwd B=%d on %d threads ..." % (batch, threads), file=sys.stderr, flush=True)
        xo.train_step_grads(x, y, sd, "M", 1)         # warm-up
        ts = []
        for _ in range(steps):
            t0 = time.time()
            xo.train_step_grads(x, y, sd, "M", 1)
            ts.append(time.time() - t0)
            print("[bench] cpu baseline step %.2f s (B=%d, %d threads)" % (ts[-1], batch, threads), file=sys.stderr, flush=True)
        return sorted(ts)[len(ts) // 2]

    t_all = timed(avail, B, 3)
    t_8 = timed(min(8, avail), B, 2)
    t_s = timed(avail, sample_B, 3)
    return {"value": round(B / t_all, 3), "unit": "clips/s", "cores": avail, "kind": "port",
            "value_8_threads": round(B / t_8, 3), "cores_8": min(8, avail), "cpu_model": _cpu_model(),
            "host_cpu_count": os.cpu_count(),
            "value_B2_sample": round(sample_B / t_s, 3),
            "sample": "oracle/x3d_oracle.py fwd+bwd, X3D-M B=%d T=%d H=W=%d fp32 = the stated workload (one step = one batch of %d "
                      "clips), median of 3 / 2 steps at %d / %d torch CPU threads; value_B2_sample: the B=%d sample earlier "
                      "rounds reported, 3 steps at %d threads" % (B, T, H, B, avail, min(8, avail), sample_B, avail)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE configs[1]: 8)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--model", default="M", choices=["S", "M", "L", "XL"],
                    help="model version (headline: M; 'L' at --size 312 with --dtype bf16 is BASELINE configs[4])")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: mixed-storage mode -- the wide tensors inside the bottlenecks stored as bf16, fp32 arithmetic "
                         "(BASELINE configs[4]; the headline metric is quoted on f32)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-overlap", action="store_true", help="multi-rank: single graph + all-reduce after the whole backward")
    ap.add_argument("--copy-input", action="store_true", help="copy the batch into the graph's input tensors every timed step "
                    "(what rounds 1-2 measured; default: the clips are resident in the graph's inputs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the second timed run with the exact fp32-MFMA backward GEMMs")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch HIP-event table (JSON) here")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # test hook (tests/test_bench_ddp_gpu.py): several ranks on ONE GPU over gloo, to exercise the
    # multi-rank control flow on a single-GPU box; the real launch is one rank per GPU over RCCL
    single_dev = os.environ.get("X3D_BENCH_SINGLE_DEVICE") == "1"
    backend = os.environ.get("X3D_BENCH_BACKEND", "nccl")
    if single_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    import x3d
    from x3dhip import ops, synthetic
    from x3dhip.trainer import Trainer

    B, T, H = args.batch, args.frames, args.size
    torch.manual_seed(0)
    mixed = args.dtype == "bf16"
    net = x3d.generate_model(args.model, n_classes=400, dropout=0.5, base_bn_splits=max(1, B // 8),
                             act_dtype=torch.bfloat16 if mixed else torch.float32).to(dev).train(True)
    tr = Trainer(net, lr=0.05, process_group=pg, world_size=world, use_graph=not args.no_graph, overlap=not args.no_overlap)
    x = synthetic.synthetic_clips(B, T, H, H, seed=1234 + rank).to(dev)
    y = synthetic.synthetic_labels(B, seed=1234 + rank).to(dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.  The synthetic clips
        are resident in HBM: after the warm-up has captured the step they live in the graph's own input tensors
        (Trainer.static_inputs -- where a device-side input pipeline writes its batches), so a timed step makes no input copy."""
        for _ in range(args.warmup):
            tr.step(x, y)
        xs, ys = x, y
        st = tr.static_inputs(x.shape) if (tr.use_graph and not args.copy_input) else None
        timed_run.input_copy = st is None and tr.use_graph      # graph mode without static inputs: one copy of x, y per step
        if st is not None:
            st[0].copy_(x)
            st[1].copy_(y)
            xs, ys = st
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_, _ = tr.step(xs, ys)
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tdt = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
            dt_ = tdt.item()
        return dt_, loss_

    dt, loss = timed_run()
    ms = 1000.0 * dt / args.steps
    value = world * B * args.steps / dt

    # the same step with the exact fp32-MFMA backward GEMMs (the switches are read per launch): reported beside the default
    exact = two_term = None
    from x3dhip import _lib, engine
    if not args.no_exact_fp32 and not mixed:          # the exact fp32-MFMA backward kernels read fp32 tensors only
        with _lib.options(dgrad_f32=1, wgrad_f32=1):  # library options are read per launch; the captured graphs are re-captured
            tr.invalidate_graphs()
            dt_e, _ = timed_run()
        exact = (world * B * args.steps / dt_e, 1000.0 * dt_e / args.steps)
        with _lib.options(bwd_terms=2):               # round 2's default: two-term operands in the backward GEMMs (~2^-16)
            tr.invalidate_graphs()
            dt_e, _ = timed_run()
        two_term = (world * B * args.steps / dt_e, 1000.0 * dt_e / args.steps)
        tr.invalidate_graphs()

    E, Ew = algorithmic_elems_M(T, H, args.model, split=True)
    nparams = sum(p.numel() for p in net.parameters())
    # 3 passes (forward, data gradient, weight gradient) over E elements; the wide ones are 2 B in bf16 mode
    step_bytes = B * 3 * (4 * (E - Ew) + (2 if mixed else 4) * Ew) + 20 * nparams
    out = {
        "metric": "clips/sec X3D-%s fwd+bwd+SGD at multigrid base shape (whole job)" % args.model,
        "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # storage, stencils, BN and the optimizer are fp32.  Pointwise GEMMs run on the bf16 MFMA with every fp32 operand split
        # into THREE bf16 terms (hi + mid + lo = all 24 significant bits; six MFMA products, fp32 accumulate: fp32-level
        # accuracy, forward AND backward; the contracting stage 1-2 forward convs and a few small layers use the fp32 MFMA
        # directly).  value_exact_fp32 is the same job with the fp32-MFMA backward kernels, value_two_term_backward with
        # round 2's two-term backward operands (~2^-16 per product).
        "dtype": ("bf16 storage of the wide bottleneck tensors (conv1 / conv2 outputs and their gradients), every other tensor "
                  "and all arithmetic f32 (pointwise GEMMs as in the f32 mode)" if mixed else
                  "f32 (pointwise GEMMs on the MFMA with fp32 operands as 3 bf16 terms, fp32 accumulate: fp32-level, fwd and bwd)"),
        "data": "synthetic",
        "config": {"workload": "X3D-%s train step B=%d/GPU T=%d H=W=%d, 400 classes, dropout 0.5, SGD momentum" % (args.model, B, T, H),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                   "storage": "bf16 wide tensors / f32" if mixed else "f32",
                   "launch": "eager" if args.no_graph else "hipGraph(fwd+bwd) + SGD",
                   # rounds 1-2 copied the batch into the graph's inputs every timed step (x: 77 MB at the
                   # headline shape, ~0.05 ms); since round 3 the synthetic clips are resident there.  --copy-input restores it.
                   "input_copy_per_step": bool(getattr(timed_run, "input_copy", False)),
                   # storage, forward GEMMs, stencils, BN: fp32.  Backward pointwise GEMMs: fp32 operands split into
                   # hi+lo bf16 (3 MFMA products, fp32 accumulate, ~2^-16 per product; parity-verified)
                   # unless X3D_DGRAD_F32 / X3D_WGRAD_F32 select the exact fp32-MFMA kernels
                   "backward_gemm": ("fp32 MFMA" if _lib.get_option("dgrad_f32") else "%d-term bf16 dgrad" % _lib.get_option("bwd_terms")) + " / " +
                                    ("fp32 MFMA" if _lib.get_option("wgrad_f32") else "%d-term bf16 wgrad" % _lib.get_option("bwd_terms")) +
                                    ("" if (_lib.get_option("dgrad_f32") or engine.cfg.no_fused_bwd)
                                     else " (stages 1-2: one fused dgrad+wgrad pass)"),
                   "loss": round(float(loss), 4)},
        "step_hbm_roofline": {"algorithmic_bytes_per_step": step_bytes,
                              "achieved_GBs": round(step_bytes / (ms * 1e-3) / 1e9, 1),
                              "frac_of_8TBs": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }

    # data-parallel exchange: who talked to whom, and what one bucket costs on its own -- so that the
    # driver's 1/2/4/8-GPU curve explains itself (the per-step exchange is overlapped with the early layers' backward)
    comm = {"rccl_ranks": world, "backend": (backend if world > 1 else None),
            "graph_mode": ("two graphs around bucket 0 (overlap)" if (world > 1 and tr.use_graph and tr._overlap())
                           else "single graph" if tr.use_graph else "eager"),
            "buckets_bytes": [4 * (b_ - a_) for a_, b_ in tr.reducer.buckets]}
    if world > 1:
        import torch.distributed as dist
        per = []
        for (a_, b_) in tr.reducer.buckets:
            buf = torch.zeros(b_ - a_, device=dev)
            for _ in range(3):
                dist.all_reduce(buf, group=pg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            e0.record()
            for _ in range(10):
                dist.all_reduce(buf, group=pg)
            e1.record()
            torch.cuda.synchronize()
            tb = torch.tensor([e0.elapsed_time(e1) / 10], device=dev, dtype=torch.float64)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            per.append(round(tb.item(), 4))
        comm["allreduce_ms_per_bucket_standalone"] = per
    out["data_parallel"] = comm
    if exact is not None:
        out["value_exact_fp32"] = round(exact[0], 2)
        out["ms_per_step_exact_fp32"] = round(exact[1], 3)
        out["value_two_term_backward"] = round(two_term[0], 2)
        out["ms_per_step_two_term_backward"] = round(two_term[1], 3)
    if not args.no_kernel_timing:
        # one instrumented eager step: HIP events around every launch, same tensors.  EVERY rank runs these steps (they
        # contain the gradient all-reduce: the collectives must match across ranks); only rank 0 records and reports.
        tr_e = tr
        tr_e.use_graph = False
        engine.cfg.side_stream = False              # serialise the weight-gradient kernels: clean per-kernel times
        for _ in range(2):          # eager warm-up: allocator + code objects outside the graph pool
            tr_e.step(x, y)
        torch.cuda.synchronize()
        if rank != 0:
            tr_e.step(x, y)
            torch.cuda.synchronize()
    if rank == 0 and not args.no_kernel_timing:
        # The eager step is host-bound (a ctypes launch costs more than most of these kernels run): with an empty queue
        # an event pair would measure the host's time between the two records.  ~60 ms of unrelated GEMM work is queued
        # first, so the whole instrumented step is enqueued behind it and the events see back-to-back device execution.
        blocker = torch.empty(8192, 8192, device=x.device).normal_()
        for _ in range(8):
            blocker @ blocker
        with KernelTimer(ops) as kt:
            tr_e.step(x, y)
        del blocker
        agg = kt.summary()
        if args.dump_launches:
            with open(args.dump_launches, "w") as f:
                json.dump(kt.launches, f)
        tot = sum(v[0] for v in agg.values())
        # dominant KERNEL (template name as rocprofv3 prints it; a family such as "pointwise forward" is four kernels)
        dom = max(kt.by_kernel.items(), key=lambda kv: kv[1][0])
        name, (tms, nbytes, cnt) = dom
        ach = nbytes / (tms * 1e-3) / 1e9 if tms > 0 else 0.0
        # HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes) and MFMA utilisation come from rocprofv3 counter
        # collections committed under profiles/ (tools/collect_traffic.py, tools/collect_pmc.py): counters cannot be read from
        # inside the run.  Each collection carries the sha of the kernel sources it was taken on; one taken on OTHER sources, or
        # of another workload, is not reported.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import glob
        import stamp
        sha = stamp.csrc_sha16()
        headline = args.model == "M" and (B, T, H) == (8, 16, 224)

        def newest(pattern):
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pattern)), reverse=True):
                try:
                    d = json.load(open(f))
                    if d.get("_meta", {}).get("csrc_sha16") == sha:
                        return f, d
                except Exception:
                    pass
            return None, None

        traffic = traffic_src = None
        tf, td = newest("*traffic_bf16_M.json" if mixed else "*traffic_f32_M.json") if headline else (None, None)
        if td is not None:
            traffic = td.get("_kernels", {}).get(name, {}).get("hbm_bytes_per_launch")
            traffic_src = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit %s)" % (os.path.relpath(tf, ROOT), td["_meta"].get("commit"))
        mfma = None
        pf, pd = newest("*pmc_sq_f32_M.json") if (headline and not mixed) else (None, None)
        if pd is not None:
            mfma = {"source": "%s (rocprofv3 --pmc pass at commit %s)" % (os.path.relpath(pf, ROOT), pd["_meta"].get("commit")),
                    "busy_matrix_pipes_of_4_per_cu": {k[7:]: round(v["derived"]["mfma_busy_per_sq_busy"], 3)
                                                      for k, v in pd.items() if k.startswith("kernel:") and
                                                      v.get("derived", {}).get("mfma_busy_per_sq_busy", 0) > 0}}
        out["mfma_utilisation"] = mfma
        out["roofline"] = {"bound": "hbm", "kernel": name, "launches_per_step": cnt,
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "avg_launch_ms": round(tms / cnt, 4),
                           "alg_bytes_per_launch": int(nbytes / cnt),
                           "share_of_step_device_time": round(tms / tot, 3),
                           "csrc_sha16": sha}
        out["kernel_roofline"] = {k: {"launches": v[2], "ms": round(v[0], 3), "avg_launch_us": round(1e3 * v[0] / v[2], 2),
                                      "achieved_GBs": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 and v[1] else None}
                                  for k, v in sorted(kt.by_kernel.items(), key=lambda kv: -kv[1][0])[:12]}
        out["kernel_breakdown_ms"] = {k: [round(v[0], 3), v[2], round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0]
                                      for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "M":
        out["cpu_baseline"] = cpu_baseline(T, H, B)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
