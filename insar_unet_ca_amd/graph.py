"""The training step of the reference's hot loop (Unet-ChannalAttention.py:342-346: zero_grad -> forward -> loss ->
backward -> optimizer.step) captured ONCE into a hipGraph and replayed.

Why: a step is ~290 (U-Net-CA) to ~900 (DeepLabV3-CA) launches whose host-side enqueue (ctypes call, descriptor fill,
torch bookkeeping) costs 4-10 ms — as long as, or longer than, the GPU work itself. A replay costs the host ~20 us and
the GPU sees the same kernels, in the same order, on the same two streams (main + weight-gradient side stream, forked
and joined inside the capture).

What makes the step capturable: every buffer of a plan is allocated once (engine.Act), the library never allocates or
synchronises, the loss and the Adam bias corrections live in device memory (optim.Adam.enable_device_step), the dropout
mask of DeepLabV3-CA is keyed by a device-side counter, and the only host decision in a step — "have the weights
changed, re-lay them out" — is always yes inside a training loop.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class GraphedTrainStep:
    """step = GraphedTrainStep(model, criterion, optimizer, x, y); loss = step(x, y) -> 0-dim device tensor (the loss of
    THIS step; reading it with .item() synchronises, as in the reference's loop :348).

    x, y fix the batch geometry. The model must be in training mode and stay there; parameters must not be re-assigned
    (model.load_state_dict copies in place and is fine; so is optimizer.load_state_dict, which restores into the
    existing state tensors — if it ever has to re-allocate them the next replay raises instead of updating freed memory). `.grad` of every parameter is the gradient of the last step."""

    def __init__(self, model: torch.nn.Module, criterion, optimizer, x: torch.Tensor, y: torch.Tensor, warmup: int = 3):
        if not x.is_cuda:
            raise _lib.InsarError("GraphedTrainStep: inputs must be ROCm tensors (no CPU fallback)")
        if not model.training:
            raise _lib.InsarError("GraphedTrainStep captures the TRAINING step: call model.train() first")
        if getattr(model, "_hooks", None) and any(k in model._hooks for k in ("on_bucket", "on_done")):
            raise _lib.InsarError("GraphedTrainStep: data-parallel hooks (RCCL collectives inside backward) are not captured; "
                                  "use the eager step under DataParallel")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.x, self.y = x.clone(), y.clone()
        if not hasattr(optimizer, "enable_device_step"):
            raise _lib.InsarError("GraphedTrainStep needs insar_unet_ca_amd.Adam (device-side step count)")
        optimizer.enable_device_step()
        # warm-up on a side stream (torch's recipe): builds the plan, the optimizer state, the pixel tables, ...
        s = torch.cuda.Stream(device=x.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.warmup_steps = max(1, warmup)
        # gradients must be (re-)created by the captured backward: first gradient -> alias of the flat buffer
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager_step()
        optimizer._dev_pending -= 1         # the capture enqueued nothing: that optimizer.step() did not happen
        # the capture itself executed nothing: parameters, optimizer state and BatchNorm buffers are those after warm-up
        self.replays = 0
        self._opt_generation = getattr(optimizer, "generation", 0)

    def _eager_step(self) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.criterion(self.model(self.x), self.y)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, x: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if y is not None:
            self.y.copy_(y, non_blocking=True)
        if getattr(self.optimizer, "generation", 0) != self._opt_generation:
            # the kernel nodes hold the addresses of the moment tensors and of the device-side step state as they were at
            # capture time; a load that re-allocated them would make the replay update freed memory
            raise _lib.InsarError("GraphedTrainStep: the optimizer's state was re-allocated after the capture "
                                  "(optimizer.load_state_dict with new shapes / first-time state); build a new GraphedTrainStep")
        self.graph.replay()
        self.optimizer._dev_pending += 1
        self.replays += 1
        return self.loss
