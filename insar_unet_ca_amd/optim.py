"""optim.Adam(lr=1e-4) of the reference's training loop (Unet-ChannalAttention.py:466,346)
as one multi-tensor HIP launch. State (`step`, `exp_avg`, `exp_avg_sq`) is kept in
torch.optim.Adam's own format, so optimizer.state_dict() interchanges with the reference's."""
from __future__ import annotations

import math
import os

import torch

from . import _lib
from ._lib import call, ptr

# elements per work-group of the Adam launch (256 threads): small enough for every CU to hold several work-groups for the
# whole launch. 31 M parameters in one buffer, stand-alone (tools/pass_bench.py): 64 Ki 177 us, 16 Ki 175, 8 Ki 166, 4 Ki 165.
CHUNK = int(os.environ.get("INSAR_ADAM_CHUNK", "8192"))


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
        self._early_tables = {}    # (plan id, stage, gradient buffer) -> (table, chunks, nchunks, params)
        self._early_done = set()   # ids of the parameters already updated inside the current backward
        self._early_stages = 0

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

    # ---- optimizer inside backward ------------------------------------------------------------------------
    def fuse_into_backward(self, model) -> "Adam":
        """Run this optimizer STAGE BY STAGE inside `loss.backward()`: as soon as a backward stage of `model` (a decoder /
        encoder block: engine.grad_groups) has enqueued its last gradient kernel, Adam updates that stage's parameters on the
        weight-gradient side stream and the stage's GEMM-layout weight copies are rebuilt right behind it — beside the rest of
        backward instead of after it (the reference's loop runs optimizer.step() after loss.backward(), :345-346; same
        arithmetic, same order per parameter, so the parameters are bit for bit those of the plain step). `step()` then only
        finishes what is left (nothing, in the steady state) and advances the step counts.
        MEASURED SLOWER on config 2 (same-box A/B 7.77 -> 8.29 ms/step, profiles/r03_adam_in_backward_ab.txt: 875 MB of optimizer
        traffic and 18 extra launches beside the dgrad chain cost more than the 0.17 ms of exposed Adam they hide), so nothing
        in this repository turns it on; it stays as a tested option for steps with a different balance.
        Contract of the opt-in: ONE backward per step() (no gradient accumulation over several backward calls: a second
        backward before step() raises), nobody reads the parameters between backward and step(), one parameter group, no
        data-parallel wrapper (there the gradients are final only after the exchange; the plain step is used)."""
        hooks = getattr(model, "_hooks", None)
        if hooks is None:
            raise _lib.InsarError("Adam.fuse_into_backward: the model has no HIP plan hooks (UNet / DeepLabV3_SingleChannel_Attn)")
        hooks["on_stage_optim"] = self._early_stage
        return self

    def _early_stage(self, plan, stage: int) -> None:
        if stage < 0:                            # backward begins
            if self._early_done:
                raise _lib.InsarError("Adam.fuse_into_backward: a second backward before optimizer.step() — the first one has "
                                      "already updated parameters; gradient accumulation needs the plain step")
            self._early_stages = 0
            return
        f = self._fast
        if (f is None or self._dev_state is not None or len(self.param_groups) != 1 or torch.cuda.is_current_stream_capturing()
                or stage >= len(plan.sink.groups)):
            return                               # not in the steady state yet (first steps build the optimizer state): plain step()
        from . import engine
        group = self.param_groups[0]
        key = (id(plan), stage, plan.sink.active)
        hit = self._early_tables.get(key)
        if hit is None:
            params = plan.sink.groups[stage]
            tensors = []
            for p in params:
                st = self.state.get(p)
                if not p.requires_grad or st is None or "exp_avg" not in st:
                    return
                tensors.append((p, plan.sink.view(p), st["exp_avg"], st["exp_avg_sq"]))
            if not tensors:
                return
            table, chunk_t, nchunks = self._table(("early",) + key, tensors)
            if len(self._early_tables) > 64:
                self._early_tables.clear()
            hit = self._early_tables[key] = (table, chunk_t, nchunks, [t[0] for t in tensors])
        table, chunk_t, nchunks, params = hit
        b1, b2 = group["betas"]
        t = float(self.state[params[0]]["step"]) + 1.0
        with plan.ctx.side_stream():             # ordered after everything this stage has enqueued on either stream
            call("insar_adam_step", ptr(table), ptr(chunk_t), nchunks, CHUNK, float(group["lr"]), float(b1), float(b2),
                 float(group["eps"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), float(self.grad_scale), _lib.stream_ptr())
            torch._C._increment_version(params)
            ws = plan.weightset.stage_sets(plan.sink.groups)[stage]
            if ws is not None and engine.PREP_SIDE:
                ws.refresh()                     # this stage's GEMM-layout copies: the next forward finds them current
        self._early_done.update(id(p) for p in params)
        self._early_stages += 1

    def _finish_early(self) -> bool:
        """step() after a backward that ran (some of) the update itself. True if nothing is left to do."""
        done, self._early_done = self._early_done, set()
        group = self.param_groups[0]
        rest = [p for p in group["params"] if p.grad is not None and id(p) not in done]
        b1, b2 = group["betas"]
        t = float(self.state[group["params"][0]]["step"]) + 1.0
        if rest:                                 # stages that did not take part (a plan change mid-way): the same kernel on them
            tensors = [(p, p.grad, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in rest]
            table, chunk_t, nchunks = self._table(tuple(x.data_ptr() for tup in tensors for x in tup), tensors)
            call("insar_adam_step", ptr(table), ptr(chunk_t), nchunks, CHUNK, float(group["lr"]), float(b1), float(b2),
                 float(group["eps"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), float(self.grad_scale), _lib.stream_ptr())
            torch._C._increment_version(rest)
        for p in group["params"]:
            st = self.state.get(p)
            if st is not None and "step" in st and (p.grad is not None or id(p) in done):
                st["step"] += 1
        return True

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
            self._early_tables.clear()
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
        params, grads, steps, table, chunk_t, nchunks, ptrs = f
        group = self.param_groups[0]
        if len(group["params"]) != len(params):
            return False
        for p, q, g, a in zip(group["params"], params, grads, ptrs):
            # same Parameter, same gradient tensor AND the same parameter storage: `p.data = ...` / load_state_dict(assign=True)
            # keep the Parameter object but move its memory, and the table holds raw addresses
            if p is not q or p.grad is not g or p.data_ptr() != a:
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
        if closure is None and self._early_done:
            self._finish_early()
            return loss
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
                    self._fast = (list(params), list(grads), list(steps), table, chunk_t, nchunks, [p.data_ptr() for p in params])
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
                self._fast = (list(params), list(grads), list(steps), table, chunk_t, nchunks, [p.data_ptr() for p in params])
        if self._dev_state is not None:
            self._dev_pending += 1
        return loss
