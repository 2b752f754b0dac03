"""Launch tapes: the fixed launch sequence of a plan's training-mode forward / backward, recorded once and replayed.

A training step of U-Net-CA is ~230 launches through the C ABI, DeepLabV3-CA ~600; the Python around each launch (descriptor
structs, attribute walks through nn.Modules, pointer look-ups) costs 10-16 us of host time, 3.7 ms per step for config 2 and
8.6 of 9.9 ms for config 5 — which makes config 5 host-bound. Every one of those launches has the same arguments step after
step (a plan owns all its buffers), except three pointers at the nn.Module boundary: the input tensor, the logits the forward
returns and the logits gradient backward receives. So: the first training-mode call of a plan runs the ordinary code (it
allocates lazily), the second and third are RECORDED (`_lib.call` appends every launch, the engine appends its stream hand-offs
and the few torch-level copies), the two recordings must agree launch for launch (same entry point, same scalar arguments,
byte-identical descriptors; the boundary pointers replaced by named slots) — and from then on the tape is replayed: one ctypes
call per launch with prebuilt arguments, the same two streams, the same order. Anything that does not fit (data-parallel
hooks, SyncBN, the per-kernel profiler, graph capture, eval mode, a recording that differs from its twin) runs the ordinary
code. `INSAR_TAPE=0` switches tapes off, `INSAR_TAPE=verify` re-records every 16th call and compares it with the tape.
Recording uses two module-level hooks (`tape.REC`, `_lib._TAPE`): one plan records at a time — two models trained from two
Python threads of one process should run with INSAR_TAPE=0.
(The reference launches eagerly, Unet-ChannalAttention.py:338-346; a hipGraph of the step replays slower than eager launches
on ROCm 7.2 — DESIGN.md — so the tape keeps eager launches and removes the host work around them.)"""
from __future__ import annotations

import sys
import ctypes as C
import os
from typing import Callable, Dict, List, Optional

import torch

from . import _lib

MODE = os.environ.get("INSAR_TAPE", "1")          # "0" off, "1" on, "verify" on + periodic re-recording
REC: Optional["Recorder"] = None                  # the recording in progress (engine appends its own ops through it)


class Recorder:
    def __init__(self, main_ptr: int, side_ptr: int, dyn: Dict[int, str]):
        self.entries: list = []
        self.main_ptr, self.side_ptr, self.dyn = main_ptr, side_ptr, dyn
        self.poison: Optional[str] = None

    # engine-side ops
    def op(self, kind: str, payload=None) -> None:
        self.entries.append((kind, payload))

    def bad(self, why: str) -> None:
        self.poison = self.poison or why

    def finish(self) -> Optional[list]:
        """Compact op list, or None if something in the recording cannot be replayed."""
        if self.poison:
            return None
        ops, in_side = [], False
        for e in self.entries:
            if len(e) == 2:
                kind, payload = e
                if kind == "side_enter":
                    in_side = True
                    ops.append((3,))
                elif kind == "side_exit":
                    in_side = False
                    ops.append((4,))
                elif kind == "join":
                    ops.append((5,))
                elif kind == "py":
                    ops.append((6, payload))
                continue
            fn, args, name = e
            if not args:
                return None
            want = self.side_ptr if in_side else self.main_ptr
            if args[-1] != want:
                self.poison = f"{name}: launched on a stream the tape does not model"
                return None
            body = list(args[:-1])
            dyn = [(i, self.dyn[a]) for i, a in enumerate(body) if isinstance(a, int) and a in self.dyn]
            for i, _ in dyn:
                body[i] = 0
            if dyn:
                ops.append((2, fn, body, 1 if in_side else 0, dyn, name))
            else:
                ops.append((1 if in_side else 0, fn, tuple(body), name))
        return ops


def _same_arg(a, b) -> bool:
    if type(a) is not type(b):
        return False
    if isinstance(a, (int, float, bytes, str)) or a is None:
        return a == b
    obj_a, obj_b = getattr(a, "_obj", None), getattr(b, "_obj", None)         # ctypes.byref(struct)
    if obj_a is not None and obj_b is not None:
        return bytes(obj_a) == bytes(obj_b)
    if isinstance(a, C.Structure):
        return bytes(a) == bytes(b)
    return a is b


def same_tape(t1: Optional[list], t2: Optional[list]) -> Optional[str]:
    """None if the two recordings describe the same launch sequence, else the first difference."""
    if t1 is None or t2 is None:
        return "a recording could not be turned into a tape"
    if len(t1) != len(t2):
        return f"{len(t1)} vs {len(t2)} ops"
    for i, (a, b) in enumerate(zip(t1, t2)):
        if a[0] != b[0]:
            return f"op {i}: kind {a[0]} vs {b[0]}"
        k = a[0]
        if k in (0, 1):
            if a[1] is not b[1] or len(a[2]) != len(b[2]) or not all(_same_arg(x, y) for x, y in zip(a[2], b[2])):
                return f"op {i} ({a[3]}): arguments differ"
        elif k == 2:
            if a[1] is not b[1] or a[3] != b[3] or a[4] != b[4] or not all(_same_arg(x, y) for x, y in zip(a[2], b[2])):
                return f"op {i} ({a[5]}): arguments differ"
    return None


def tape_py(fn: Callable[[], None]) -> None:
    """Run a torch-level operation of a plan (a small copy between plan-owned tensors) and, while recording, put it on the
    tape. The closure must only touch tensors the plan owns."""
    fn()
    if REC is not None:
        REC.op("py", fn)


def tape_live(fn: Callable[[], None]) -> None:
    """Run `fn` — plan code whose launches depend on state a tape cannot see (WeightSet.refresh: are the GEMM copies of the
    weights stale?) — with recording suspended, and put `fn` ITSELF on the tape: a replay calls it at the same point of the
    launch sequence, on whatever stream is current there, and it decides afresh every time."""
    global REC
    rec = REC
    if rec is None:
        fn()
        return
    REC, _lib._TAPE = None, None
    try:
        fn()
    finally:
        REC, _lib._TAPE = rec, rec.entries
    rec.op("py", fn)


class PlanTape:
    """Mixin of engine.UNetPlan / deeplab.DeepLabPlan: `forward` / `backward` dispatch between the ordinary code
    (`_forward_eager`, `_backward_eager`) and the tape of that code."""

    def _tape_setup(self) -> None:
        self._tapes: Dict[tuple, dict] = {}
        self._tape_params = list(self.net.parameters())

    # what must not change under a tape (cheap to evaluate per call)
    def _tape_key(self, which: str) -> tuple:
        from . import engine
        bn = tuple((m.momentum, m.eps) for m in self.bn_modules)
        # module-level switches the launch code reads at call time (tests flip them with monkeypatch): part of the key too
        flags = (engine.BSTAT_FUSE, engine.BSTAT_C64, engine.COEF_SIMPLE, engine.COEF_FUSE, engine.SPLIT_COEF, engine.POOL_FUSE,
                 engine.OUTC_FUSE, engine.OUTC_WGRAD_FUSE, engine.PREP_SIDE, engine.SMALL_WGRAD_MAIN, engine.SMALL_WGRAD_FUSE, engine.WGRAD_LATE, engine.WGRAD_ROWS, engine.WGRAD_X, engine.WGRAD_Y, engine.WGRAD_K, engine.FLAT_PP, engine.FLAT2, engine.FLAT2_KMAX, engine.FLAT2_PERSIST,
                 engine.FLAT_PERSIST, engine.FLAT_ROWS, engine.IGEMM_PP, engine.WGRAD_FILL, engine.WGRAD_FILL_SMALL, engine.WGRAD_FILL_T, engine.WGRAD_FILL_DL, engine.WGRAD_GRID_CAP,
                 getattr(sys.modules.get(__package__ + ".deeplab"), "GATE_FUSE", None), getattr(sys.modules.get(__package__ + ".deeplab"), "GATE_STATS", None))
        return (which, self.sink.active if which == "b" else 0, hash(bn), hash(flags))

    def _tape_fingerprint(self) -> int:
        """Storage addresses a tape bakes into its prebuilt arguments: every parameter of the plan and every BatchNorm buffer.
        The ordinary code re-reads them per call (p.data = ..., load_state_dict(assign=True), module.to(), a swapped
        running_mean all work there); a tape whose fingerprint no longer matches is dropped and recorded again."""
        ptrs = [p.data_ptr() for p in self._tape_params]
        for m in self.bn_modules:
            ptrs.append(m.running_mean.data_ptr() if m.running_mean is not None else 0)
            ptrs.append(m.running_var.data_ptr() if m.running_var is not None else 0)
            ptrs.append(m.num_batches_tracked.data_ptr() if m.num_batches_tracked is not None else 0)
        return hash(tuple(ptrs))

    def _tape_allowed(self, training: bool, extra_ok: bool = True) -> bool:
        from . import engine
        return (MODE != "0" and training and extra_ok and engine.PROFILER is None and self.ctx.side is not None
                and not self.net._hooks.get("sync_bn") and not torch.cuda.is_current_stream_capturing())

    def _run(self, key: tuple, eager: Callable, slots: Dict[str, int], dyn_ptrs: Dict[int, str], dyn_after=None):
        """eager(): the ordinary code, returns its result. Returns (result or None, replayed?). dyn_ptrs: boundary pointers
        known before the call (pointer -> slot name); dyn_after(result): those only known afterwards (the logits tensor)."""
        st = self._tapes.setdefault(key, {"state": 0, "tape": None, "calls": 0, "fp": None})
        st["calls"] += 1
        fp = self._tape_fingerprint()
        if st["fp"] != fp:                        # storage moved under the tape (or first call): start over
            if st["state"] not in (0, -1):
                st["moved"] = st.get("moved", 0) + 1
            st["state"], st["tape"], st["fp"] = (0 if st.get("moved", 0) < 8 else -1), None, fp
        if st["state"] == 3:
            if MODE == "verify" and st["calls"] % 16 == 0:
                out, rec = self._record(eager, dyn_ptrs, dyn_after)
                why = same_tape(st["tape"], rec)
                if why:
                    raise _lib.InsarError(f"launch tape {key}: the live launch sequence departed from the tape: {why}")
                return out, False
            self._replay(st["tape"], slots)
            return None, True
        if st["state"] in (0, -1):               # first call (lazy allocations) / tape given up
            if st["state"] == 0:
                st["state"] = 1
            return eager(), False
        out, rec = self._record(eager, dyn_ptrs, dyn_after)
        if st["state"] == 1:
            st["tape"], st["state"] = rec, 2
        else:
            why = same_tape(st["tape"], rec)
            st["state"] = 3 if why is None else -1
            st["why"] = why
            if why is not None:
                st["tape"] = None
        return out, False

    def _record(self, eager: Callable, dyn_ptrs: Dict[int, str], dyn_after=None):
        global REC
        side = self.ctx.side.cuda_stream
        rec = Recorder(_lib.stream_ptr(), side, dict(dyn_ptrs))
        REC, _lib._TAPE = rec, rec.entries
        try:
            out = eager()
        finally:
            REC, _lib._TAPE = None, None
        if dyn_after is not None:
            rec.dyn.update(dyn_after(out))
        return out, rec.finish()

    def _replay(self, ops: list, slots: Dict[str, int]) -> None:
        ctx = self.ctx
        main = _lib.stream_ptr()
        side = ctx.side.cuda_stream
        prev = None                               # torch's current stream while a side section is open
        try:
            for op in ops:
                k = op[0]
                if k == 0:
                    rc = op[1](*op[2], main)
                elif k == 1:
                    rc = op[1](*op[2], side)
                elif k == 2:
                    body = list(op[2])
                    for i, slot in op[4]:
                        body[i] = slots[slot]
                    rc = op[1](*body, side if op[3] else main)
                elif k == 3:
                    prev = ctx._side_enter()
                    continue
                elif k == 4:
                    ctx._side_exit(prev)
                    prev = None
                    continue
                elif k == 5:
                    ctx.join_side()
                    continue
                else:
                    op[1]()
                    continue
                if rc != 0:
                    raise _lib.InsarError(f"{op[-1]} failed ({rc}) during tape replay: {_lib.load().insar_last_error().decode(errors='replace')}")
        finally:
            if prev is not None:                  # a launch failed inside a side section: restore the caller's stream
                ctx._side_exit(prev)

    def tape_report(self) -> Dict[tuple, str]:
        names = {0: "not yet", 1: "recording", 2: "recording", 3: "replaying", -1: "given up"}
        return {k: names[v["state"]] + (f" ({v.get('why')})" if v.get("why") else "") + (f", {len(v['tape'])} ops" if v.get("tape") else "")
                for k, v in self._tapes.items()}
