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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            b1, b2 = group["betas"]
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
        return loss
