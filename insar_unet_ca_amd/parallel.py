"""Data-parallel training over the GPUs of one MI355X node: one process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI). The reference is single-device
(Unet-ChannalAttention.py:406-419); this inserts the gradient exchange between
`loss.backward()` (:345) and `optimizer.step()` (:346).

Gradients of one replica live in a single flat fp32 buffer laid out in backward-completion
order (engine.GradSink), so the exchange is a handful of large bucketed collectives issued
from inside the backward pass as soon as a bucket's last weight-gradient kernel has been
enqueued: RCCL runs them on its own stream, overlapped with the remaining backward kernels.
Two exchange schemes (same bytes on the links: an all-reduce IS a reduce-scatter + an all-gather):

* default: mean all-reduce of every bucket, then the ordinary `Adam` on every rank (each rank
  updates all 31.26 M parameters: 875 MB of optimizer traffic per rank and step);
* `DataParallel(..., shard_optimizer=True)` + `ShardedAdam`: mean reduce-scatter of every bucket,
  Adam on this rank's 1/world slice of each bucket (optimizer traffic and state divided by world),
  all-gather of the updated parameters. Parameters then live in one flat
  buffer with the gradient buffer's layout; the `nn.Parameter`s are views of it, so state_dict /
  checkpoints are unchanged.

BatchNorm statistics stay per replica (standard DDP semantics); `sync_buffers()` averages
the running statistics over the ranks (call it before validation / checkpointing).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import engine


class BucketReducer:
    """Averages contiguous slices of a flat gradient buffer across ranks, asynchronously."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.works: List = []
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)

    def reset(self) -> None:
        """Forget collectives of a backward pass that did not reach `finish` (it raised half-way)."""
        for work, _ in self.works:
            try:
                work.wait()
            except Exception:
                pass
        self.works.clear()

    def reduce_slice(self, flat: torch.Tensor, begin: int, end: int) -> None:
        if end <= begin:
            return
        view = flat[begin:end]
        if self.backend == "nccl":
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        else:   # gloo has no AVG
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), view))

    def scatter_slice(self, flat: torch.Tensor, begin: int, end: int, out: torch.Tensor) -> None:
        """out (1/world of the slice) = this rank's part of the mean of flat[begin:end] over the ranks."""
        view = flat[begin:end]
        if self.backend == "nccl":
            self.works.append((dist.reduce_scatter_tensor(out, view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        else:
            self.works.append((dist.reduce_scatter_tensor(out, view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), out))

    def finish(self) -> None:
        for work, view in self.works:
            work.wait()                 # nccl: the current stream waits for the collective; no host sync
            if view is not None:
                view.div_(self.world)
        self.works.clear()


def plan_buckets(sizes: List[int], min_elems: int) -> List[int]:
    """Given per-stage gradient sizes in completion order, return the stage indices after which a
    bucket closes (stages are merged until a bucket holds at least `min_elems` elements)."""
    closes, acc = [], 0
    for i, n in enumerate(sizes):
        acc += n
        if acc >= min_elems or i == len(sizes) - 1:
            closes.append(i)
            acc = 0
    return closes


def bucket_bounds(sizes: List[int], min_elems: int) -> List[Tuple[int, int]]:
    """[begin, end) element ranges of the buckets `plan_buckets` forms over the flat buffer."""
    ends, acc = [], 0
    for n in sizes:
        acc += n
        ends.append(acc)
    out, begin = [], 0
    for i in plan_buckets(sizes, min_elems):
        out.append((begin, ends[i]))
        begin = ends[i]
    return out


class ShardedBuckets:
    """This rank's 1/world slice of every bucket: its parameters (`p`: a VIEW of the flat parameter buffer, so whatever
    changes the parameters in place — `load_state_dict` on resume, a manual re-initialisation — is what the next
    optimizer step starts from, and the all-gather runs in place), the averaged gradients (`g`) and the Adam moments
    (`m`, `v`). Bucket b covers flat[begin_b:end_b]; rank r owns flat[begin_b + r*n_b : begin_b + (r+1)*n_b],
    n_b = (end_b - begin_b) / world (engine.GROUP_ALIGN makes every bucket divisible into 16-byte aligned shards for
    world <= 16)."""

    def __init__(self, flat_p: torch.Tensor, bounds: List[Tuple[int, int]], world: int, rank: int):
        self.bounds, self.world, self.rank = bounds, world, rank
        self.p, self.g, self.m, self.v = [], [], [], []
        for b, e in bounds:
            if (e - b) % (4 * world):
                raise ValueError(f"bucket of {e - b} elements does not split into {world} 16-byte aligned shards")
            n = (e - b) // world
            own = flat_p[b + rank * n: b + (rank + 1) * n]
            self.p.append(own)
            self.g.append(torch.zeros_like(own))
            self.m.append(torch.zeros_like(own))
            self.v.append(torch.zeros_like(own))

    def rows(self):
        return list(zip(self.p, self.g, self.m, self.v))


class DataParallel(torch.nn.Module):
    """DDP-style wrapper for `insar_unet_ca_amd.UNet`.

    - parameters and BN buffers are broadcast from rank 0 at construction;
    - every backward pass exchanges the gradients in buckets of >= `bucket_mb` MiB, overlapped with the rest of
      backward: mean all-reduce (default) or, with `shard_optimizer=True`, mean reduce-scatter into this rank's
      shard (use `ShardedAdam` as the optimizer then; `p.grad` stays None);
    - `sync_bn=True`: BatchNorm statistics over the GLOBAL batch (SyncBN; default: per replica, standard DDP semantics);
    - `forward` and `state_dict` delegate to the wrapped module (no `module.` prefix games:
      use `.module.state_dict()` for reference-compatible checkpoints).
    """

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: float = 16.0,
                 shard_optimizer: bool = False, sync_bn: bool = False):
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before wrapping a model in DataParallel")
        self.module = module
        self.pg = process_group
        self.reducer = BucketReducer(process_group)
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.min_elems = int(bucket_mb * (1 << 20) / 4)
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, src=0, group=process_group)
        self._stage = 0
        self._begin = 0
        self._bucket = 0
        self.sharded: Optional[ShardedBuckets] = None
        self.flat_p: Optional[torch.Tensor] = None
        self._gathers: dict = {}               # bucket index -> all-gather in flight (ShardedAdam(defer_gather=True))
        self._stage_bucket: List[int] = []     # backward stage -> bucket that holds its parameters
        self.gather_wait_events: List = []     # bench.py: HIP events around each forward-side wait for a parameter bucket
        self.record_exposed = False            # bench.py: HIP events around the wait for the gradient exchange
        self.exposed_events: List = []
        if shard_optimizer:
            self._flatten_parameters()
        module._hooks["on_begin"] = self._on_begin
        module._hooks["on_bucket"] = self._on_bucket
        module._hooks["on_done"] = self._on_done
        if shard_optimizer:
            module._hooks["grad_mode"] = "none"
            module._hooks["before_forward"] = self._before_forward
            # plans that re-lay their weights stage by stage (engine.UNetPlan) wait for a bucket only when the first layer
            # that needs it is about to run: the encoder starts while the decoder's buckets are still arriving
            module._hooks["param_waits"] = self.wait_stage
        if sync_bn:
            # synchronised BatchNorm (U-Net-CA plan): global-batch statistics, i.e. the reference's single-device batch
            # semantics under data parallelism, for two tiny all-reduces per BatchNorm layer and step
            module._hooks["sync_bn"] = (process_group, self.world)

    def forward(self, *args, **kwargs):
        if not getattr(self.module, "per_stage_param_waits", False):
            self._params_ready()
        return self.module(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        self._params_ready()
        return super().state_dict(*args, **kwargs)

    # ---- sharded-optimizer layout ---------------------------------------------------------------------
    def _flatten_parameters(self) -> None:
        """Move every parameter into ONE flat fp32 buffer with the gradient buffer's layout (engine.flat_layout over
        engine.grad_groups); the nn.Parameters become views of it. Cached plans hold raw pointers: dropped."""
        groups = engine.grad_groups(self.module)
        params = [p for g in groups for p in g]
        if len({id(p) for p in params}) != len(list(self.module.parameters())):
            raise RuntimeError("shard_optimizer: the flat layout does not cover every parameter of the module")
        offs, sizes, total = engine.flat_layout(groups)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        with torch.no_grad():
            for p, o in zip(params, offs):
                if p.dtype != torch.float32:
                    raise RuntimeError("shard_optimizer: float32 master parameters only")
                flat[o:o + p.numel()].copy_(p.reshape(-1))
                p.data = flat[o:o + p.numel()].view(p.shape)
        for m in self.module.modules():
            if hasattr(m, "_plans") and hasattr(m._plans, "clear"):
                m._plans.clear()
        self.flat_p = flat
        self.flat_params = params
        self.sharded = ShardedBuckets(flat, bucket_bounds(sizes, self.min_elems), self.world, self.rank)
        closes = plan_buckets(sizes, self.min_elems)
        self._stage_bucket = [next(b for b, c in enumerate(closes) if st <= c) for st in range(len(sizes))]

    def gather_parameters(self, wait: bool = False) -> None:
        """All-gather the updated shards into the flat parameter buffer (after ShardedAdam's kernel), in place: every
        rank's shard already sits at its slot of the bucket. The buckets are issued in the order the NEXT forward pass needs
        them (the flat layout is backward-completion order, so the last bucket holds the first encoder levels) and nothing
        waits here: `_params_ready` (run by the module's forward before its first launch, or by `wait=True`) makes the
        compute streams wait, so the collectives overlap whatever the training loop does between optimizer.step() and the
        next forward (zero_grad, the input copy, logging)."""
        nccl = self.reducer.backend == "nccl"
        self._params_ready()                          # (a step() without a forward in between: nothing may still be in flight)
        for k in reversed(range(len(self.sharded.bounds))):
            (b, e), own = self.sharded.bounds[k], self.sharded.p[k]
            src = own if nccl else own.clone()        # gloo copies the input into the output slot: keep them distinct
            self._gathers[k] = dist.all_gather_into_tensor(self.flat_p[b:e], src, group=self.pg, async_op=True)
        torch._C._increment_version(self.flat_params)      # cached GEMM-layout copies of the weights are now stale
        if wait:
            self._params_ready()

    def params_ready(self) -> None:
        """Order the current stream after the parameter all-gathers of the last ShardedAdam.step(defer_gather=True)."""
        self._params_ready()

    def _params_ready(self, plan=None) -> None:
        for k in sorted(self._gathers, reverse=True):      # issue order
            self._gathers[k].wait()  # nccl: the current stream waits for the collective; no host sync
        self._gathers.clear()

    def _before_forward(self, plan) -> None:
        """before the first launch of a forward pass: plans without stage-wise weight re-layout wait for every bucket"""
        if not getattr(plan, "per_stage_param_waits", False):
            self._params_ready()

    def wait_stage(self, stage: int) -> None:
        """Make the CURRENT stream wait for the all-gather of the bucket that holds backward stage `stage` (and, issue
        order being forward-need order, nothing else: buckets the forward pass needs later keep flying)."""
        if not self._gathers:
            return
        k = self._stage_bucket[stage]
        w = self._gathers.pop(k, None)
        if w is None:
            return
        if self.record_exposed and self.flat_p is not None and self.flat_p.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            w.wait()
            e1.record()
            self.gather_wait_events.append((e0, e1))
        else:
            w.wait()

    @torch.no_grad()
    def sync_buffers(self) -> None:
        """Average the floating-point BatchNorm buffers (running_mean / running_var) over the ranks and take rank 0's
        integer buffers (num_batches_tracked is equal anyway): replicas then validate / checkpoint identically."""
        for b in self.module.buffers():
            if b.is_floating_point():
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg)
                b.div_(self.world)
            else:
                dist.broadcast(b, src=0, group=self.pg)

    # ---- called by the engine from inside backward --------------------------------------------------------
    def _on_begin(self, plan) -> None:
        # a backward that raised mid-way must not leave its bookkeeping to the next step
        self.reducer.reset()
        self._stage = self._begin = self._bucket = 0

    def _exchange(self, plan, begin: int, end: int) -> None:
        if self.sharded is not None:
            b, e = self.sharded.bounds[self._bucket]
            if (b, e) != (begin, end):
                raise RuntimeError(f"bucket {self._bucket}: plan closes [{begin},{end}), optimizer shards [{b},{e})")
            self.reducer.scatter_slice(plan.sink.flat(), begin, end, self.sharded.g[self._bucket])
        else:
            self.reducer.reduce_slice(plan.sink.flat(), begin, end)
        self._bucket += 1

    def _on_bucket(self, plan, tag) -> None:
        """after each backward stage's gradients have been enqueued"""
        ends, closes = plan.stage_ends, plan.bucket_closes(self.min_elems)
        if self._stage in closes:
            end = ends[self._stage]
            # The bucket's weight gradients were computed on the side stream, its BN / SE gradients on the main
            # stream. Issuing the collective from the side stream (which first waits for the main stream's
            # position) orders it after both WITHOUT stalling the main stream's dgrad chain.
            with plan.ctx.side_stream():
                self._exchange(plan, self._begin, end)
            self._begin = end
        self._stage += 1

    def _on_done(self, plan) -> None:
        total = plan.sink.flat().numel()
        if self._begin < total:
            plan.ctx.join_side()
            self._exchange(plan, self._begin, total)
        if self.record_exposed and plan.sink.flat().is_cuda:
            # from the end of backward's own kernels on the main stream to the point where the reduced gradients are usable
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.reducer.finish()
            e1.record()
            self.exposed_events.append((e0, e1))
        else:
            self.reducer.finish()
        self._stage = self._begin = self._bucket = 0


def _hip_adam_rows(rows, lr, b1, b2, eps, bc1, bc2_sqrt) -> None:
    """One multi-tensor launch of the Adam kernel over (p, g, m, v) rows (the product path; no CPU fallback)."""
    from . import _lib
    from .optim import CHUNK
    dev = rows[0][0].device
    if dev.type != "cuda":
        raise _lib.InsarError("ShardedAdam HIP path: parameters must live on a ROCm device (no CPU fallback)")
    table, chunks = [], []
    for ti, (p, g, m, v) in enumerate(rows):
        table.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()])
        chunks += [[ti, ci] for ci in range((p.numel() + CHUNK - 1) // CHUNK)]
    key = tuple(r[0] for r in table)
    cache = _hip_adam_rows.cache
    if key not in cache:
        cache.clear()
        cache[key] = (torch.tensor(table, dtype=torch.int64).to(dev), torch.tensor(chunks, dtype=torch.int32).to(dev))
    t, c = cache[key]
    _lib.call("insar_adam_step", _lib.ptr(t), _lib.ptr(c), c.shape[0], CHUNK, float(lr), float(b1), float(b2), float(eps),
              float(bc1), float(bc2_sqrt), 1.0, _lib.stream_ptr())


_hip_adam_rows.cache = {}


class ShardedAdam:
    """optim.Adam(lr=1e-4) (Unet-ChannalAttention.py:466,346) on this rank's shard of every bucket, followed by the
    all-gather of the updated parameters. Same arithmetic per element as `optim.Adam` (the same kernel), so with equal
    reduced gradients the parameters equal the all-reduce path's bit for bit. `kernel` replaces the HIP launch in the
    CPU (gloo) tests of the exchange logic; the product path leaves it None."""

    def __init__(self, dp: DataParallel, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 kernel: Optional[Callable] = None, defer_gather: bool = False):
        """defer_gather: return from step() with the parameter all-gathers still in flight; the wrapped module's next
        forward (or DataParallel.state_dict / params_ready()) makes the compute stream wait for them. Off by default:
        code that reads `dp.module` parameters right after step() then needs no extra call."""
        if dp.sharded is None:
            raise RuntimeError("ShardedAdam needs DataParallel(..., shard_optimizer=True)")
        self.dp, self.lr, self.betas, self.eps = dp, lr, betas, eps
        self.defer_gather = defer_gather
        self.kernel = kernel or _hip_adam_rows
        self.steps = 0

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.dp.flat_params:
            p.grad = None

    @torch.no_grad()
    def step(self) -> None:
        self.steps += 1
        b1, b2 = self.betas
        self.kernel(self.dp.sharded.rows(), self.lr, b1, b2, self.eps, 1.0 - b1 ** self.steps,
                    math.sqrt(1.0 - b2 ** self.steps))
        self.dp.gather_parameters(wait=not self.defer_gather)

    def state_dict(self) -> dict:
        sh = self.dp.sharded
        return {"steps": self.steps, "rank": sh.rank, "world": sh.world, "bounds": list(sh.bounds),
                "exp_avg": [t.clone() for t in sh.m], "exp_avg_sq": [t.clone() for t in sh.v]}

    def load_state_dict(self, sd: dict) -> None:
        sh = self.dp.sharded
        if (sd["rank"], sd["world"], list(sd["bounds"])) != (sh.rank, sh.world, list(sh.bounds)):
            raise RuntimeError("ShardedAdam: checkpoint was written for another rank / world size / bucket plan")
        self.steps = sd["steps"]
        for dst, src in zip(sh.m, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(sh.v, sd["exp_avg_sq"]):
            dst.copy_(src)
