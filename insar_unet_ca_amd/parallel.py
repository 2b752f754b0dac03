"""Data-parallel training over the GPUs of one MI355X node: one process per GPU,
`torch.distributed` backend "nccl" (= RCCL over xGMI). The reference is single-device
(Unet-ChannalAttention.py:406-419); this inserts the gradient exchange between
`loss.backward()` (:345) and `optimizer.step()` (:346).

Gradients of one replica live in a single flat fp32 buffer laid out in backward-completion
order (engine.GradSink), so the exchange is a handful of large bucketed all-reduces issued
from inside the backward pass as soon as a bucket's last weight-gradient kernel has been
enqueued: RCCL runs them on its own stream, overlapped with the remaining backward kernels.
BatchNorm statistics stay per replica (standard DDP semantics).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class BucketReducer:
    """Averages contiguous slices of a flat gradient buffer across ranks, asynchronously."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.works: List = []
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)

    def reduce_slice(self, flat: torch.Tensor, begin: int, end: int) -> None:
        if end <= begin:
            return
        view = flat[begin:end]
        if self.backend == "nccl":
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        else:   # gloo has no AVG
            self.works.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), view))

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


class DataParallel(torch.nn.Module):
    """DDP-style wrapper for `insar_unet_ca_amd.UNet`.

    - parameters and BN buffers are broadcast from rank 0 at construction;
    - every backward pass all-reduces (mean) the gradients in buckets of >= `bucket_mb` MiB,
      overlapped with the rest of backward;
    - `forward` and `state_dict` delegate to the wrapped module (no `module.` prefix games:
      use `.module.state_dict()` for reference-compatible checkpoints).
    """

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: float = 16.0):
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before wrapping a model in DataParallel")
        self.module = module
        self.pg = process_group
        self.reducer = BucketReducer(process_group)
        self.min_elems = int(bucket_mb * (1 << 20) / 4)
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, src=0, group=process_group)
        self._stage = 0
        self._begin = 0
        module._hooks["on_bucket"] = self._on_bucket
        module._hooks["on_done"] = self._on_done

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # called by the engine from inside backward, after each stage's gradients have been enqueued
    def _on_bucket(self, plan, tag) -> None:
        ends, closes = plan.stage_ends, plan.bucket_closes(self.min_elems)
        if self._stage in closes:
            end = ends[self._stage]
            # The bucket's weight gradients were computed on the side stream, its BN / SE gradients on the main
            # stream. Issuing the all-reduce from the side stream (which first waits for the main stream's
            # position) orders the collective after both WITHOUT stalling the main stream's dgrad chain.
            with plan.ctx.side_stream():
                self.reducer.reduce_slice(plan.sink.flat(), self._begin, end)
            self._begin = end
        self._stage += 1

    def _on_done(self, plan) -> None:
        total = plan.sink.flat().numel()
        if self._begin < total:
            plan.ctx.join_side()
            self.reducer.reduce_slice(plan.sink.flat(), self._begin, total)
        self.reducer.finish()
        self._stage = 0
        self._begin = 0
