"""optim.Adam(lr=1e-4) of the reference's training loop (Unet-ChannalAttention.py:466,346)
as one multi-tensor HIP launch. State (`step`, `exp_avg`, `exp_avg_sq`) is kept in
torch.optim.Adam's own format, so optimizer.state_dict() interchanges with the reference's."""
from __future__ import annotations

import math

import torch

from . import _lib
from ._lib import call, ptr

CHUNK = 65536


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise _lib.InsarError("Adam HIP path: weight_decay=0, amsgrad=False, maximize=False only")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)
        self._tables = {}
        self.grad_scale = 1.0      # multiplies every gradient inside the kernel (DP pre-scaling)
        self._dev_state = None     # float[4] on the device: step count + bias corrections (enable_device_step)
        self._dev_pending = 0      # steps taken on the device that state['step'] has not been told about yet
        self._fast = None          # (params, grads, step tensors, table, chunks, nchunks) of the last full step
        self.generation = 0        # bumped whenever optimizer state had to be re-allocated (see load_state_dict)

    # ---- device-side step count (hipGraph capture) ---------------------------------------------------------
    def enable_device_step(self) -> None:
        """Keep the step count and the bias corrections in device memory (insar_adam_step_dev): the launch arguments of
        `step()` then never change, which is what a captured hipGraph of the training step needs. The arithmetic is the
        same (bias corrections computed in double, rounded to fp32: bitwise the eager path's parameters). Requires every
        parameter to share one step count; `state_dict()` still reports torch.optim.Adam's per-parameter `step`."""
        if self._dev_state is not None:
            return
        steps, dev = set(), None
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p, {})
                steps.add(float(st["step"]) if "step" in st else 0.0)
                dev = p.device
        if len(steps) > 1:
            raise _lib.InsarError("Adam.enable_device_step: parameters have different step counts")
        if len({tuple(g["betas"]) for g in self.param_groups}) > 1:
            raise _lib.InsarError("Adam.enable_device_step: one (beta1, beta2) for all parameter groups")
        t = steps.pop() if steps else 0.0
        b1, b2 = self.param_groups[0]["betas"]
        self._dev_state = torch.tensor([t, 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), 0.0], dtype=torch.float32).to(dev)

    def _sync_host_steps(self) -> None:
        if self._dev_pending:
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st is not None and "step" in st:
                        st["step"] += self._dev_pending
            self._dev_pending = 0

    def state_dict(self):
        self._sync_host_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Restores IN PLACE wherever the optimizer already holds state of the same shape: the moment tensors, the step
        counters and the device-side step state keep their addresses, so cached pointer tables and a captured hipGraph
        (graph.GraphedTrainStep bakes those addresses into its kernel nodes) stay valid and see the restored values.
        State that has to be re-allocated (first load, changed shapes) bumps `generation`, which GraphedTrainStep checks."""
        self._sync_host_steps()
        old = {p: dict(st) for p, st in self.state.items()}
        super().load_state_dict(state_dict)          # builds fresh state tensors
        in_place = True
        for p, st in self.state.items():
            prev = old.get(p)
            for k in ("exp_avg", "exp_avg_sq", "step"):
                new = st.get(k)
                keep = prev.get(k) if prev else None
                if (torch.is_tensor(new) and torch.is_tensor(keep) and keep.shape == new.shape and keep.dtype == new.dtype
                        and keep.device == new.device):
                    keep.copy_(new)
                    st[k] = keep
                elif new is not None:
                    in_place = False
        if set(old) - set(self.state):
            in_place = False
        if not in_place:
            self._fast = None
            self._tables.clear()
            self.generation += 1
        if self._dev_state is not None:
            self._dev_pending = 0
            keep, self._dev_state = self._dev_state, None
            self.enable_device_step()                # validates the loaded step counts, builds the new values
            keep.copy_(self._dev_state)              # ... which go into the tensor a captured graph already points at
            self._dev_state = keep

    def _table(self, key, tensors):
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        dev = tensors[0][0].device
        rows, chunks = [], []
        for ti, (p, g, m, v) in enumerate(tensors):
            rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()])
            for ci in range((p.numel() + CHUNK - 1) // CHUNK):
                chunks.append([ti, ci])
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        chunk_t = torch.tensor(chunks, dtype=torch.int32).to(dev)
        if len(self._tables) > 8:
            self._tables.clear()
        self._tables[key] = (table, chunk_t, len(chunks))
        return self._tables[key]

    def _fast_step(self) -> bool:
        """The steady-state step: same parameters, same gradient tensors (the flat-buffer views the model hands out every
        step) and optimizer state as the last full step -> reuse its pointer table; the per-parameter bookkeeping of
        torch.optim.Adam (_init_group, 400 data_ptr calls) costs 0.7-1.2 ms of host time per step otherwise."""
        f = self._fast
        if f is None or len(self.param_groups) != 1:
            return False
        params, grads, steps, table, chunk_t, nchunks = f
        group = self.param_groups[0]
        if len(group["params"]) != len(params):
            return False
        for p, q, g in zip(group["params"], params, grads):
            if p is not q or p.grad is not g:
                return False
        b1, b2 = group["betas"]
        if self._dev_state is not None:
            call("insar_adam_step_dev", ptr(table), ptr(chunk_t), nchunks, CHUNK, float(group["lr"]), float(b1), float(b2),
                 float(group["eps"]), ptr(self._dev_state), float(self.grad_scale), _lib.stream_ptr())
        else:
            for st in steps:
                st += 1
            t = float(steps[0])
            call("insar_adam_step", ptr(table), ptr(chunk_t), nchunks, CHUNK, float(group["lr"]), float(b1), float(b2),
                 float(group["eps"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), float(self.grad_scale), _lib.stream_ptr())
        torch._C._increment_version(params)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if closure is None and self._fast_step():
            if self._dev_state is not None:
                self._dev_pending += 1
            return loss
        self._fast = None
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            b1, b2 = group["betas"]
            if self._dev_state is not None:
                tensors = []
                for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
                    if not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous():
                        raise _lib.InsarError("Adam HIP path: contiguous float32 ROCm parameters and gradients only")
                    tensors.append((p, g, m, v))
                key = tuple(x.data_ptr() for tup in tensors for x in tup)
                table, chunk_t, nchunks = self._table(key, tensors)
                call("insar_adam_step_dev", ptr(table), ptr(chunk_t), nchunks, CHUNK, float(group["lr"]), float(b1), float(b2),
                     float(group["eps"]), ptr(self._dev_state), float(self.grad_scale), _lib.stream_ptr())
                torch._C._increment_version(params)
                if len(self.param_groups) == 1:
                    self._fast = (list(params), list(grads), list(steps), table, chunk_t, nchunks)
                continue
            by_step = {}
            for p, g, m, v, st in zip(params, grads, exp_avgs, exp_avg_sqs, steps):
                if not p.is_cuda:
                    raise _lib.InsarError("Adam HIP path: parameters must live on a ROCm device (no CPU fallback)")
                if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous():
                    raise _lib.InsarError("Adam HIP path: contiguous float32 parameters and gradients only")
                st += 1
                by_step.setdefault(float(st), []).append((p, g, m, v))
            for t, tensors in by_step.items():
                key = tuple(x.data_ptr() for tup in tensors for x in tup)
                table, chunk_t, nchunks = self._table(key, tensors)
                bc1 = 1.0 - b1 ** t
                bc2_sqrt = math.sqrt(1.0 - b2 ** t)
                call("insar_adam_step", ptr(table), ptr(chunk_t), nchunks, CHUNK, float(group["lr"]), float(b1), float(b2),
                     float(group["eps"]), bc1, bc2_sqrt, float(self.grad_scale), _lib.stream_ptr())
                torch._C._increment_version([p for p, _, _, _ in tensors])
            if len(self.param_groups) == 1 and len(by_step) == 1 and len(params) == len(group["params"]):
                self._fast = (list(params), list(grads), list(steps), table, chunk_t, nchunks)
        if self._dev_state is not None:
            self._dev_pending += 1
        return loss
