"""Host-side execution plans for the U-Net-CA hot path.

A plan owns every device buffer of one (batch, height, width, dtype) configuration
— activations are padded NHWC (`Act`), sized once and reused every step — and turns
`UNet.forward` / `loss.backward()` (Unet-ChannalAttention.py:127-163, :345) into a fixed
sequence of launches through the C ABI (`_lib.call`) on torch's current HIP stream.
PyTorch only provides device memory, streams and the autograd edge at the module boundary.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib, tape
from ._lib import InsarAct, InsarBnFinalize, InsarBnSeBwd, InsarIgemm, InsarSeFwd, InsarWgrad, call, ptr
from .tape import tape_py

BN_ROW_PIX = 128        # igemm M-tile (rows of one stats slab row)
WG_BKP = 64             # wgrad pixels per K step


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class Act:
    """A channel slice [c_off, c_off+c_len) of a padded NHWC buffer [B][H+2][W+2][C]."""

    def __init__(self, buf: torch.Tensor, B: int, H: int, W: int, Ctot: int, c_off: int, c_len: int):
        self.buf, self.B, self.H, self.W, self.C, self.c_off, self.c_len = buf, B, H, W, Ctot, c_off, c_len
        self.code = _lib.dtype_code(buf.dtype)
        self.desc = InsarAct(buf.data_ptr(), B, H, W, Ctot, c_off, c_len, self.code, 0)
        self.ref = C.byref(self.desc)

    @staticmethod
    def alloc(B, H, W, Cn, dtype, device) -> "Act":
        nbytes = B * (H + 2) * (W + 2) * Cn * (2 if dtype == torch.bfloat16 else 4)
        if nbytes >= 2 ** 32:
            raise _lib.InsarError(f"activation buffer of {nbytes} bytes exceeds the 4 GiB addressing budget")
        return Act(torch.zeros((B, H + 2, W + 2, Cn), dtype=dtype, device=device), B, H, W, Cn, 0, Cn)

    def slice(self, c_off: int, c_len: int) -> "Act":
        return Act(self.buf, self.B, self.H, self.W, self.C, self.c_off + c_off, c_len)

    def nchw(self) -> torch.Tensor:
        """Interior as a float32 NCHW tensor (debug / tests only)."""
        v = self.buf[:, 1:-1, 1:-1, self.c_off:self.c_off + self.c_len]
        return v.permute(0, 3, 1, 2).float().contiguous()


class Ctx:
    """Per-(device, dtype) shared scratch: reduction temporaries, wgrad slabs, pixel tables."""

    def __init__(self, device: torch.device, dtype: torch.dtype):
        self.device, self.dtype = device, dtype
        self.code = _lib.dtype_code(dtype)
        self.esize = 2 if dtype == torch.bfloat16 else 4
        self.ch = 16 // self.esize
        self._colsum_tmp: Dict[int, torch.Tensor] = {}
        # Weight gradients (wgrad GEMM + slab folds) depend on nothing but a layer's input and its dy, and
        # nothing in backward depends on them: they go to a second HIP stream so that they overlap the
        # HBM-bound BN/ReLU backward passes and the next dgrad on the main stream. INSAR_SIDE_STREAM=0 disables.
        # INSAR_SIDE_PRIORITY: HIP stream priority of the side stream (lower number = higher priority; diagnostic)
        self.side = (torch.cuda.Stream(device=device, priority=int(os.environ.get("INSAR_SIDE_PRIORITY", "0")))
                     if device.type == "cuda" and os.environ.get("INSAR_SIDE_STREAM", "1") != "0" else None)
        # INSAR_SIDE_CU_MASK = <n>[:stride] (diagnostic, measured in round 3): the side stream confined to n compute units
        # (hipExtStreamCreateWithCUMask; bits 0..n-1 of the mask, or every stride-th bit)
        mask = os.environ.get("INSAR_SIDE_CU_MASK")
        if self.side is not None and mask:
            self.side = _cu_masked_stream(device, mask)
        self._side_busy = False
        self._wgrad_part: Optional[torch.Tensor] = None
        self._wgrad_fold: Optional[torch.Tensor] = None
        self._tables: Dict[tuple, torch.Tensor] = {}
        self._consts: Dict[tuple, torch.Tensor] = {}

    # ---- small helpers ------------------------------------------------------------------------
    def f32(self, *shape) -> torch.Tensor:
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def const(self, value: float, n: int) -> torch.Tensor:
        key = (value, n)
        if key not in self._consts:
            self._consts[key] = torch.full((n,), value, dtype=torch.float32, device=self.device)
        return self._consts[key]

    @contextlib.contextmanager
    def side_stream(self):
        """Run the enclosed launches on the side stream, ordered after everything enqueued so far on the
        current stream. Per-kernel timing passes (PROFILER) stay on one stream."""
        if self.side is None or (PROFILER is not None and PROFILER.alone):
            yield
            return
        prev = self._side_enter()
        try:
            yield
        finally:
            self._side_exit(prev)

    def _side_enter(self):
        """(also what a launch tape replays) side stream waits for the current stream's position and becomes current"""
        if tape.REC is not None:
            tape.REC.op("side_enter")
        prev = torch.cuda.current_stream()
        self.side.wait_stream(prev)
        torch.cuda.set_stream(self.side)
        return prev

    def _side_exit(self, prev) -> None:
        torch.cuda.set_stream(prev)
        self._side_busy = True
        if tape.REC is not None:
            tape.REC.op("side_exit")

    def join_side(self) -> None:
        """Order the current stream after the side stream's work (end of backward, before a gradient bucket
        is handed to the all-reduce)."""
        if tape.REC is not None:
            tape.REC.op("join")
        if self._side_busy:
            torch.cuda.current_stream().wait_stream(self.side)
            self._side_busy = False

    def colsum(self, part: torch.Tensor, out: torch.Tensor, segments: int, rows: int, cols: int,
               accumulate: bool = False, ld: Optional[int] = None, part_off: int = 0) -> None:
        """out[s][c] (+)= sum_r part[s][r][c]; ld: row stride of `part` when only its first `cols` columns (from float
        offset `part_off`) are wanted — a gradient summed straight into its view of the flat gradient buffer."""
        s = _lib.stream_ptr()
        tmp = self._colsum_tmp.get(s)
        need = ((rows + 127) // 128) * segments * cols if rows > 256 else 0
        if tmp is None or need > tmp.numel():
            tmp = self._colsum_tmp[s] = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=self.device)
        if ld is None:
            call("insar_colsum", ptr(part), ptr(out), segments, rows, cols, int(accumulate), ptr(tmp), tmp.numel(), s)
        else:
            call("insar_colsum_ld", ptr(part) + 4 * part_off, ptr(out), segments, rows, cols, ld, int(accumulate), ptr(tmp), tmp.numel(), s)

    def colsum_into_grads(self, part: torch.Tensor, rows: int, views: List[torch.Tensor]) -> None:
        """Fold part[rows][sum of the views' sizes] into consecutive parameter-gradient views (outc's weight and bias): one
        launch when the views are adjacent in the flat buffer, one per view otherwise; no staging tensor, no copy."""
        total = sum(v.numel() for v in views)
        adjacent = all(views[i + 1].data_ptr() == views[i].data_ptr() + 4 * views[i].numel() for i in range(len(views) - 1))
        if adjacent:
            self.colsum(part, views[0], 1, rows, total)
            return
        off = 0
        for v in views:
            self.colsum(part, v, 1, rows, v.numel(), ld=total, part_off=off)
            off += v.numel()

    def wgrad_part(self, floats: int) -> torch.Tensor:
        if self._wgrad_part is None or self._wgrad_part.numel() < floats:
            self._wgrad_part = torch.empty(floats, dtype=torch.float32, device=self.device)
        return self._wgrad_part

    def wgrad_finish(self, part: torch.Tensor, grad: torch.Tensor, nsplit: int, ntaps: int, cout: int, cin: int,
                     layout: int) -> None:
        """Fold the split-K slabs (two stages when there are many) and write the torch-layout gradient."""
        s = _lib.stream_ptr()
        slab = ntaps * cout * cin
        if nsplit > 48:           # up to 48 slabs the re-layout kernel folds them itself (four reads in flight per thread)
            group = 8 if nsplit <= 96 else 16
            groups = (nsplit + group - 1) // group
            if self._wgrad_fold is None or self._wgrad_fold.numel() < groups * slab:
                self._wgrad_fold = torch.empty(groups * slab, dtype=torch.float32, device=self.device)
            call("insar_wgrad_fold", ptr(part), ptr(self._wgrad_fold), slab, nsplit, group, s)
            part, nsplit = self._wgrad_fold, groups
        call("insar_wgrad_reduce", ptr(part), ptr(grad), nsplit, ntaps, cout, cin, layout, 0, s)

    def pixel_table(self, B, H, W, s, Hb, Wb, tail) -> torch.Tensor:
        key = (B, H, W, s, Hb, Wb, tail)
        if key not in self._tables:
            mpad = _round_up(B * H * W, WG_BKP)
            t = torch.empty(mpad, dtype=torch.int32, device=self.device)
            call("insar_pixel_table", ptr(t), mpad, B, H, W, s, Hb, Wb, tail, _lib.stream_ptr())
            self._tables[key] = t
        return self._tables[key]


def _cu_masked_stream(device: torch.device, spec: str):
    """A HIP stream whose kernels may only use some compute units (hipExtStreamCreateWithCUMask), wrapped for torch."""
    n, _, stride = spec.partition(":")
    n, stride = int(n), int(stride or 1)
    words = [0] * 8                      # 256 CUs
    for i in range(n):
        b = (i * stride) % 256 + (i * stride) // 256
        words[b // 32] |= 1 << (b % 32)
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p()
    arr = (C.c_uint32 * 8)(*words)
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, arr)
    if rc != 0:
        raise _lib.InsarError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(stream.value, device=device)


def _taps_table(self, B: int, Ho: int, Wo: int, s: int, Hb: int, Wb: int, taps) -> torch.Tensor:
    """Per-tap pixel tables [ntaps][Mpad] for insar_wgrad (strided / dilated convolutions): entry = padded pixel index of
    (n, ho*s + dy, wo*s + dx) in a (Hb, Wb) buffer, or 0 (the zero halo corner) outside it."""
    key = ("taps", B, Ho, Wo, s, Hb, Wb, tuple(taps))
    if key not in self._tables:
        mpad = _round_up(B * Ho * Wo, WG_BKP)
        t = torch.empty(len(taps) * mpad, dtype=torch.int32, device=self.device)
        dy = (C.c_int8 * 12)(*([a for a, _ in taps] + [0] * (12 - len(taps))))
        dx = (C.c_int8 * 12)(*([b for _, b in taps] + [0] * (12 - len(taps))))
        call("insar_pixel_table_taps", ptr(t), mpad, B, Ho, Wo, s, Hb, Wb, len(taps), C.addressof(dy), C.addressof(dx),
             _lib.stream_ptr())
        self._tables[key] = t
    return self._tables[key]


Ctx.taps_table = _taps_table


class GemmWeight:
    """GEMM-operand copies ([tap][n][k], compute dtype) of a Conv2d / ConvTranspose2d weight: the forward
    form and the transposed / flipped input-gradient form. Refreshed when the fp32 master parameter changes
    (tensor version + storage); a WeightSet refreshes all weights of a plan in ONE launch."""

    def __init__(self, ctx: Ctx, param: torch.nn.Parameter, kind: str):
        self.ctx, self.param, self.kind = ctx, param, kind
        if kind == "conv3":           # any Conv2d (Co, Ci, k, k): T = k*k taps in raster order
            co, ci = param.shape[0], param.shape[1]
            T = param.shape[2] * param.shape[3]
            self._spec = {"fwd": (T, co, ci, 1, ci * T, T), "dgrad": (T, ci, co, 1, T, ci * T)}
        else:       # convT (Ci, Co, 2, 2): forward rows n = q*Co + co
            ci, co = param.shape[0], param.shape[1]
            self._spec = {"fwd": (4, co, ci, 1, 4, co * 4), "dgrad": (4, ci, co, 1, co * 4, 4)}
        self._buf = {k: torch.empty(v[:3], dtype=ctx.dtype, device=ctx.device) for k, v in self._spec.items()}
        self._key = {"fwd": None, "dgrad": None}

    def key(self):
        return (self.param._version, self.param.data_ptr())

    def master(self) -> torch.Tensor:
        w = self.param.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise _lib.InsarError("weights must be contiguous float32 parameters")
        return w

    def _get(self, which: str) -> torch.Tensor:
        if self._key[which] != self.key():
            if tape.REC is not None:       # a per-weight re-layout depends on state the tape does not see: stay eager
                tape.REC.bad("weight re-layout inside a recording")
            T, N, K, st, sn, sk = self._spec[which]
            call("insar_weight_prep", ptr(self.master()), ptr(self._buf[which]), self.ctx.code, T, N, K, st, sn, sk,
                 _lib.stream_ptr())
            self._key[which] = self.key()
        return self._buf[which]

    def fwd(self) -> torch.Tensor:
        return self._get("fwd")

    def dgrad(self) -> torch.Tensor:
        return self._get("dgrad")


class WeightSet:
    """All GemmWeights of a plan: one batched re-layout launch per optimizer step."""

    def __init__(self, ctx: Ctx, weights: List[GemmWeight]):
        self.ctx, self.weights = ctx, weights
        self._jobs = None
        self._ptrs = None
        self._total = 0
        self._by_stage = None

    def stage_sets(self, groups) -> List["WeightSet"]:
        """One WeightSet per backward stage (the GEMM weights whose master parameter belongs to that stage): an optimizer
        that runs stage by stage inside backward (optim.Adam.fuse_into_backward) re-lays a stage's weights right behind its
        update, and the next forward finds nothing stale."""
        if self._by_stage is None:
            owner = {id(p): i for i, g in enumerate(groups) for p in g}
            buckets = [[] for _ in groups]
            for w in self.weights:
                buckets[owner[id(w.param)]].append(w)
            self._by_stage = [WeightSet(self.ctx, b) if b else None for b in buckets]
        return self._by_stage

    def stale(self) -> bool:
        return any(w._key["fwd"] != w.key() or w._key["dgrad"] != w.key() for w in self.weights)

    def refresh(self) -> None:
        """Re-lay the GEMM copies if any master moved. On a launch tape this is ONE live op (tape.tape_live): the check
        runs on every replay exactly as it does here, and nothing it launches is baked into the tape — a recording made
        while no weight was stale (several forwards before the first optimizer step) must not lose the re-layout."""
        tape.tape_live(self._refresh_now)

    def _refresh_now(self) -> None:
        if not self.stale():
            return
        ptrs = tuple(w.param.data_ptr() for w in self.weights)
        if self._jobs is None or ptrs != self._ptrs:
            rows, tile0 = [], 0
            for w in self.weights:
                # master in[a][b][T]: Conv2d (Co,Ci,3,3): out_ab = forward form, out_ba = dgrad form;
                # ConvTranspose2d (Ci,Co,2,2): out_ab = dgrad form, out_ba = forward form
                a, b = w.param.shape[0], w.param.shape[1]
                T = w.param.shape[2] * w.param.shape[3]
                oab, oba = (w._buf["fwd"], w._buf["dgrad"]) if w.kind == "conv3" else (w._buf["dgrad"], w._buf["fwd"])
                rows.append([w.master().data_ptr(), oab.data_ptr(), oba.data_ptr(), a, b, T, tile0, w.ctx.code])
                tile0 += ((a + 31) // 32) * ((b + 31) // 32)
            self._jobs = torch.tensor(rows, dtype=torch.int64).to(self.ctx.device)
            self._ptrs, self._total = ptrs, tile0
        call("insar_weight_prep_pair_batch", ptr(self._jobs), self._jobs.shape[0], self._total, _lib.stream_ptr())
        for w in self.weights:
            w._key["fwd"] = w._key["dgrad"] = w.key()


class KernelTimer:
    """Optional HIP-event timing of the GEMM-class launches (bench.py's roofline leg). Events are
    recorded on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self, alone: bool = True):
        # alone: every launch on ONE stream (no weight-gradient overlap, split-K that fills the chip): per-kernel
        # durations. Otherwise the launch configuration of an ordinary step is kept (side stream, half-chip split-K)
        # and the events of a side-stream kernel are recorded on the side stream.
        self.alone = alone
        self.elapsed = 0.0
        self.records: Dict[str, list] = {}

    def run(self, tag: str, flops: float, fn, nbytes: float = 0.0) -> None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.setdefault(tag, []).append((e0, e1, flops, nbytes))

    def summary(self) -> Dict[str, dict]:
        torch.cuda.synchronize()
        out = {}
        for tag, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            fl = sum(r[2] for r in recs)
            out[tag] = {"launches": len(recs), "ms": ms, "flops": fl, "bytes": sum(r[3] for r in recs),
                        "avg_us": 1e3 * ms / max(len(recs), 1), "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
        return out


PROFILER: Optional[KernelTimer] = None
# Measured (B=16, 256x256, bf16): splitting the backward coefficient launch (batch fold on the side stream, k1/k2 folded
# inside the apply pass) is 0.5 % SLOWER (9.40 vs 9.355 ms/step): the side stream already fills the main stream's
# latency gaps in backward, which is throughput-bound. Kept as an option (bitwise equal, tested), off by default.
SPLIT_COEF = os.environ.get("INSAR_SPLIT_COEF", "0") == "1"
# Measured: issuing a unit's weight gradient AFTER its input-gradient GEMM ("after") is 1.2 % slower (9.34 vs 9.23 ms).
POOL_FUSE = os.environ.get("INSAR_POOL_FUSE", "1") != "0"        # diagnostic: 0 routes the max-pool gradient with insar_maxpool2_bwd
SMALL_WGRAD_FUSE = os.environ.get("INSAR_SMALL_WGRAD_FUSE", "1") != "0"   # ... with the unit's BatchNorm-backward apply pass evaluated inside it (dy never written)
SMALL_WGRAD_MAIN = os.environ.get("INSAR_SMALL_WGRAD_MAIN", "1") != "0"   # the first layer's weight gradient (last launch of backward) on the main stream
PREP_SIDE = os.environ.get("INSAR_PREP_SIDE", "1") != "0"         # diagnostic: 0 re-lays the weights out on the main stream
OUTC_FUSE = os.environ.get("INSAR_OUTC_FUSE", "1") != "0"        # diagnostic: 0 materialises the gradient of outc's input
WGRAD_GRID_CAP = int(os.environ.get("INSAR_WGRAD_GRID_CAP", "200"))    # 8-wave weight-gradient launches beside the dgrad chain: at most this many work-groups (0 = off)
COEF_FUSE = os.environ.get("INSAR_COEF_FUSE", "1") != "0"        # diagnostic: 0 = two launches for the coefficient stages of every unit
COEF_SIMPLE = os.environ.get("INSAR_COEF_SIMPLE", "1") != "0"    # units without an SE gate: the channel-parallel one-launch kernel (0 = the ticket kernel)
OUTC_WGRAD_FUSE = os.environ.get("INSAR_OUTC_WGRAD_FUSE", "1") != "0"   # diagnostic: 0 = outc's weight gradient in its own pass over y
WGRAD_FILL_T = float(os.environ.get("INSAR_WGRAD_FILL_T", "0.7"))   # the same for the transposed convs' per-tap weight gradient
WGRAD_FILL_DL = float(os.environ.get("INSAR_WGRAD_FILL_DL", "0.5"))  # ... and for DeepLabV3-CA's per-tap weight gradients (1x1 / dilated / strided convs: most of its side stream; config 5 same-box 0.5 9.32, 0.6 9.42, 0.7 9.51, 0.85 9.60 ms/step)
WGRAD_FILL_ALONE = float(os.environ.get("INSAR_WGRAD_FILL_ALONE", "1.0"))   # ... and when it has the GPU to itself
WGRAD_FILL_SMALL = float(os.environ.get("INSAR_WGRAD_FILL_SMALL", os.environ.get("INSAR_WGRAD_FILL", "0.5")))   # ... for the 64- / 128-channel layers of the 256^2 / 128^2 levels (csrc/wgrad3.hip), which are HBM-heavy where the deep layers are MFMA-bound (swept 0.3 - 1.0: nothing beats the deep layers' value, profiles/r04_flat2.txt item 12)
WGRAD_FILL = float(os.environ.get("INSAR_WGRAD_FILL", "0.5"))     # share of the work-group slots a side-stream weight gradient aims at. Re-swept whenever the main queue's kernels change: 0.5 in round 2, 0.6 with round 3's row tiles (profiles/r03_row_tiles.txt), 0.5 again with the two-work-group flat kernel (same box, 120-step runs: 0.4 6.85, 0.45 6.80, 0.5 6.74, 0.55 6.79, 0.6 6.79, 0.7 6.80 ms/step; profiles/r04_flat2.txt)
WGRAD_ROWS = os.environ.get("INSAR_WGRAD_ROWS", "1") != "0"      # diagnostic: 0 = per-tap weight-gradient kernel everywhere
WGRAD_K = os.environ.get("INSAR_WGRAD_K", "0") != "0"            # 1 = wgrad3k.hip (pixel slices per wave) for the 64 / 128-channel layers named by INSAR_WGRAD_K_TILES. Off: faster alone on two tile shapes, but in the step 6.90 vs 6.915 ms in three interleaved rounds (its KS slabs per work-group cost more fold traffic than the loop saves)
# (Cin x Cout) tiles that take wgrad3k.hip. Measured (profiles/r04_wgrad3k.txt): alone 128 x 64 182.7 -> 155.8 us, 128 x 128 77.8 -> 72.4,
# 64 x 128 50.9 -> 48.3, 64 x 64 102 -> 128 (two ring slots only: slower); in the step every choice is within noise of the
# 128-tile kernel except 64 x 64 (+0.07 ms): the shallow levels are not bound by the K loop's structure
WGRAD_K_TILES = set(os.environ.get("INSAR_WGRAD_K_TILES", "128x64,128x128").split(","))
WGRAD_X = os.environ.get("INSAR_WGRAD_X", "1") != "0"            # diagnostic: 0 = the 128 x 128 row-of-taps kernel (wgrad3.hip) also where the 256 x 128 six-phase kernel (wgrad3x.hip) applies; slabs bit for bit equal at equal nsplit
WGRAD_Y = int(os.environ.get("INSAR_WGRAD_Y", "0"))            # ... on 128 x 128 tiles by 4-wave work-groups, two per CU (csrc/wgrad3y.hip), bit 0: instead of wgrad3x wherever both channel counts are multiples of 128, bit 1: instead of wgrad3<128,128> (the 128 -> 128 layers). Measured slower both ways (profiles/r04_wgrad3y.txt)
FLAT_PP = int(os.environ.get("INSAR_FLAT_PP", "1"))               # flat 3x3 kernel: ping-pong tap steps (0 = plain loop; bitwise equal)
FLAT_PERSIST = int(os.environ.get("INSAR_FLAT_PERSIST", "2"))     # flat 3x3 kernel: one work-group per CU walking its tiles (bitwise equal): 1 = forward launches, 2 = all (default since round 3: with the BatchNorm-backward sums carried over a work-group's tiles the input-gradient launches gain too, same-box 7.55 -> 7.49 ms/step), 0 = off
IGEMM_PP = int(os.environ.get("INSAR_IGEMM_PP", "1"))            # 256 x 256 tiles: ping-pong K loop (0 = the plain two-slab loop; bitwise equal)
# BatchNorm statistics slabs with more rows than this get a pre-fold launch (insar_colsum_partial) before insar_bn_finalize;
# up to it the finalize launch folds the slab itself (16 row lanes x 8 rows in flight: 1024 rows are 8 round trips, cheaper
# than the extra launch on the forward chain)
STAT_PREFOLD_ROWS = int(os.environ.get("INSAR_STAT_PREFOLD_ROWS", "1024"))
# 3x3 convs the per-tap kernel would run on 256 x 128 / 256 x 64 tiles (the 32^2 and 16^2 levels, the 128-column layers of the
# 64^2 / 128^2 levels) go to the flat kernel's row-tile geometry instead (the dx taps share one staged A tile, same tile count:
# 12-24 % faster per launch, tools/gemm_bench.py --what rows -> profiles/r03_row_tiles.txt). 0 = per-tap kernel everywhere.
FLAT_ROWS = os.environ.get("INSAR_FLAT_ROWS", "1") != "0"
# flat 3x3 kernel of the 256^2 / 128^2 levels as two co-resident 4-wave work-groups per CU (csrc/conv3x3_flat2.hip: one group's
# prologue / epilogue under the other's K loop, 128 x 64 wave tiles), persistent grids of two groups per CU (FLAT_PERSIST):
# 1 = its row tiles where the grid is 128 / 256 pixels wide (M / 256 tiles: the persistent groups get equal shares), the flat
# geometry elsewhere; 3 = the flat geometry everywhere; 2 = flat geometry, one work-group per tile; 0 = the 8-wave kernel
# (results differ by fp32 summation order only). Same box, 150-step runs, 0 -> 3: 6.963 -> 6.855 ms/step (profiles/r04_flat2.txt)
FLAT2 = int(os.environ.get("INSAR_FLAT2", "1"))
FLAT2_PERSIST = int(os.environ.get("INSAR_FLAT2_PERSIST", "1"))   # ... its persistent grids (two work-groups per CU walking their tiles, BatchNorm sums carried): 1 = forward launches only, 2 = all, 0 = never. Beside the weight-gradient stream, whose 156 KB work-groups own whole CUs, a static two-per-CU grid waits for CUs it does not get: one work-group per tile there (same box, 120-step runs: 2 -> 1 6.917 -> 6.850 ms/step, 0 6.848)
FLAT2_KMAX = int(os.environ.get("INSAR_FLAT2_KMAX", "512"))       # ... its row tiles instead of the per-tap kernel's 256 x 256 tiles up to this many input channels (0 = never)
BSTAT_FUSE = os.environ.get("INSAR_BSTAT_FUSE", "1") != "0"      # diagnostic: 0 = BatchNorm-backward sums always in a pass of their own
BSTAT_C64 = os.environ.get("INSAR_BSTAT_C64", "1") != "0"       # the 64 -> 64 kernel's variant of it. Alone it costs more than it saves (8-byte y loads from the accumulator layout: 89 -> 140 us per launch against the 43 us reduce pass it replaces), in the step it wins (same-box A/B 7.70 -> 7.61 ms: one launch less on the dgrad chain beside the weight-gradient stream)
WGRAD_LATE = os.environ.get("INSAR_WGRAD_ORDER", "before") == "after"      # diagnostic / tuning switch, see ConvBN.backward


_TAPS3 = [(r - 1, s - 1) for r in range(3) for s in range(3)]
_TAPS3_DGRAD = [(1 - r, 1 - s) for r in range(3) for s in range(3)]
_TAPS2 = [(a, b) for a in range(2) for b in range(2)]


def _igemm(x: Act, y: Act, w, N: int, Ho: int, Wo: int, stride: int, taps, mode: int,
           bias: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, oob: bool = False,
           add: Optional[Act] = None, out_stride: int = 1, out_off=(0, 0), bstat=None, gate: Optional[Act] = None) -> None:
    """w: tensor, or a raw device pointer (a tap slice of a GEMM-layout weight). bstat = (y Act, scale, shift): the stats
    slab receives the BatchNorm-backward sums of the unit that consumes this GEMM's output (InsarBstat). gate: the result is
    stored as zero where this tensor (the output's layout) is <= 0."""
    d = InsarIgemm()
    if bstat is not None:
        d.bstat.y, d.bstat.scale, d.bstat.shift = bstat[0].buf.data_ptr(), ptr(bstat[1]), ptr(bstat[2])
    d.x, d.y = x.desc, y.desc
    d.w, d.bias, d.stats = (w if isinstance(w, int) else ptr(w)), ptr(bias), ptr(stats)
    d.N, d.Ho, d.Wo, d.stride, d.ntaps, d.mode = N, Ho, Wo, stride, len(taps), mode
    d.flags = (_lib.IGEMM_OOB_ZERO if oob else 0) | (_lib.IGEMM_PINGPONG if IGEMM_PP else 0)
    d.out_stride, d.out_oy, d.out_ox = out_stride, out_off[0], out_off[1]
    if add is not None:
        if add.C != y.C or add.c_off != y.c_off or add.buf.dtype != y.buf.dtype or add.buf.shape != y.buf.shape:
            raise _lib.InsarError("igemm: `add` must have the output's buffer layout")
        d.add = add.buf.data_ptr()
    if gate is not None:
        if gate.C != y.C or gate.c_off != y.c_off or gate.buf.dtype != y.buf.dtype or gate.buf.shape != y.buf.shape:
            raise _lib.InsarError("igemm: `gate` must have the output's buffer layout")
        d.gate = gate.buf.data_ptr()
    for i, (dy, dx) in enumerate(taps):
        d.dy[i], d.dx[i] = dy, dx
    if PROFILER is not None:
        flops = 2.0 * x.B * Ho * Wo * N * x.c_len * len(taps)
        bm = call("insar_igemm_tile_rows", x.B * Ho * Wo, N)
        bn = call("insar_igemm_tile_cols_dt", x.B * Ho * Wo, N, x.code)
        if oob and x.code == _lib.F32:
            bn = 64                       # fp32 out-of-bounds variants: 64-column tiles only
        tag = "igemm_kernel<%s, %d, %d, %d%s>%s" % ("float" if x.code == _lib.F32 else "bf16_t", bm, bn,
                                                    3 if bm == 256 and bn < 256 else 2, ", oob" if oob else "",
                                                    " +bstat" if bstat is not None else "")
        es = 2 if x.code == _lib.BF16 else 4      # operands each read once, output written once
        nbytes = es * (x.B * x.H * x.W * x.c_len + y.B * y.H * y.W * y.c_len + len(taps) * N * x.c_len)
        PROFILER.run(tag, flops, lambda: call("insar_igemm", C.byref(d), _lib.stream_ptr()), nbytes)
        return
    call("insar_igemm", C.byref(d), _lib.stream_ptr())


def _flat_persist(flip: int) -> bool:
    """Persistent work-groups for the flat kernel: 2 = every launch (default), 1 = forward launches only (round 2's default:
    beside the side stream's weight gradients a static tile assignment then lost what it gained; since the input-gradient
    launches also carry the consumer's BatchNorm-backward sums over their tiles it wins there too), 0 = never."""
    return FLAT_PERSIST == 2 or (FLAT_PERSIST == 1 and not (flip & 1))


def _flat_flags(flip: int, x: Act) -> int:
    """flip bits of a flat-geometry launch: bit 2 persistent work-groups, bit 5 the two-work-group kernel (bf16)."""
    two = FLAT2 and x.code == _lib.BF16
    rows = two and FLAT2 == 1 and call("insar_conv3x3_flat2_rows_ok", x.ref, 64)
    persist = _flat2_persist(flip) if two else _flat_persist(flip)
    return (32 if two else 0) | (8 if rows else 0) | (4 if (persist and not (two and FLAT2 == 2)) else 0)


def _flat2_persist(flip: int) -> bool:
    return FLAT2_PERSIST == 2 or (FLAT2_PERSIST == 1 and not (flip & 1))


def _rows_flags(x: Act, N: int, K: Optional[int] = None, flip: int = 0) -> int:
    """flip bits (8 | 16) of the flat kernel's row-tile geometry for a 3x3 conv of x's grid to N channels, or 0 where the
    per-tap kernel keeps the layer: fp32, grids the geometry does not cover, and wherever the per-tap kernel runs its
    256 x 256 ping-pong tiles on more than FLAT2_KMAX channels (it wins there). 64-column tiles (bit 4) where 128-column ones
    would give fewer than 256. K: the GEMM's K where it is not x's channel count (input gradients)."""
    if not FLAT_ROWS or x.code != _lib.BF16 or not call("insar_conv3x3_flat_rows_ok", x.ref, N):
        return 0
    M = x.B * x.H * x.W
    # enough 128-column tiles for two work-groups per CU: the two-work-group kernel's row tiles, persistent
    two = (FLAT2 == 1 and (N % 128) == 0 and (M // 256) * (N // 128) >= 2 * torch.cuda.get_device_properties(x.buf.device).multi_processor_count
           and call("insar_conv3x3_flat2_rows_ok", x.ref, N))
    if call("insar_igemm_tile_cols_dt", M, N, x.code) == 256:
        # ... also against the 256 x 256 ping-pong tiles where K is short (256 -> 256 at 64^2 73 / 82 -> 65 / 65 us, 128 -> 256 at
        # 128^2 141 -> 133; at K = 512 the two are equal alone and the step reads 0.01 ms better: profiles/r04_flat2.txt, item 8)
        return (8 | 32 | (4 if _flat2_persist(flip) else 0)) if (two and (x.c_len if K is None else K) <= FLAT2_KMAX) else 0
    narrow = (N % 128) != 0 or (M // 256) * (N // 128) < 256
    if two:
        return 8 | 32 | (4 if _flat2_persist(flip) else 0)
    return 8 | (16 if narrow else 0)


def _conv3x3_flat(x: Act, y: Act, w: torch.Tensor, flip: int, stats: Optional[torch.Tensor], bstat=None, geo: int = 0) -> None:
    flags = flip | (2 if (FLAT_PP or geo) else 0) | (0 if geo else _flat_flags(flip, x)) | geo
    if bstat is not None:
        bs = _lib.InsarBstat(bstat[0].buf.data_ptr(), ptr(bstat[1]), ptr(bstat[2]))
        fn = lambda: call("insar_conv3x3_flat_bstat", x.ref, y.ref, ptr(w), flags, ptr(stats), C.byref(bs), _lib.stream_ptr())
    else:
        fn = lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(w), flags, ptr(stats), _lib.stream_ptr())
    if PROFILER is not None:
        flops = 2.0 * x.B * x.H * x.W * y.c_len * x.c_len * 9
        tag = "conv3x3_flat_kernel<%s, %d>%s%s" % ("float" if x.code == _lib.F32 else "bf16_t",
                                                   128 if (y.c_len % 128 == 0 and not (geo & 16)) else 64,
                                                   (" row tiles" + (" dilated" if (geo >> 8) & 15 else "")) if geo else "", " +bstat" if bstat is not None else "")
        if flags & 32:
            tag = "conv3x3_flat2_kernel<%d, %s, %d>" % (128 if y.c_len % 128 == 0 else 64, "true" if bstat is not None else "false", (flags >> 3) & 1)
        # algorithmic bytes of the launch: the input and the output once, the nine weight slabs once (+ the consumer's y for +bstat)
        es = 2 if x.code == _lib.BF16 else 4
        nbytes = float(x.B * x.H * x.W * (x.c_len + y.c_len * (2 if bstat is not None else 1)) * es + 9 * x.c_len * y.c_len * es)
        PROFILER.run(tag, flops, fn, nbytes)
        return
    fn()


def _conv3x3_c64(x: Act, y: Act, w: torch.Tensor, flip: int, stats: Optional[torch.Tensor], bstat=None) -> None:
    if bstat is not None:
        bs = _lib.InsarBstat(bstat[0].buf.data_ptr(), ptr(bstat[1]), ptr(bstat[2]))
        fn = lambda: call("insar_conv3x3_c64_bstat", x.ref, y.ref, ptr(w), flip, ptr(stats), C.byref(bs), _lib.stream_ptr())
    else:
        fn = lambda: call("insar_conv3x3_c64", x.ref, y.ref, ptr(w), flip, ptr(stats), _lib.stream_ptr())
    if PROFILER is not None:
        PROFILER.run("conv3x3_c64_kernel<2>" + (" +bstat" if bstat is not None else ""), 2.0 * x.B * x.H * x.W * 64 * 64 * 9, fn)
        return
    fn()


def _wgrad_tiles(cin: int, cout: int, code: int):
    """(Cin, Cout) tile of the weight-gradient kernel (csrc/wgrad.hip: insar_wgrad_tile)."""
    pair = call("insar_wgrad_tile_pair", cin, cout, code)
    return pair >> 16, pair & 0xffff


def _launch_wgrad(d: InsarWgrad, M: int, cin: int, cout: int, ntaps: int, code: int) -> None:
    if PROFILER is not None:
        tm, tn = _wgrad_tiles(cin, cout, code)
        tag = "wgrad_kernel<%s, %d, %d, %d>" % ("float" if code == _lib.F32 else "bf16_t", tm, tn,
                                                  8 if max(tm, tn) == 256 or (code == _lib.F32 and tm == 128) else 4)
        es = 2 if code == _lib.BF16 else 4
        nbytes = es * M * (cin + cout) * 1.0 + 4.0 * d.nsplit * ntaps * cout * cin
        PROFILER.run(tag, 2.0 * M * cin * cout * ntaps, lambda: call("insar_wgrad", C.byref(d), _lib.stream_ptr()), nbytes)
        return
    call("insar_wgrad", C.byref(d), _lib.stream_ptr())


def _side_fill(ctx: "Ctx", bf16_fill: float) -> float:
    """Share of the work-group slots a side-stream weight gradient aims at. bf16: half (the main stream's kernels are
    partly HBM- / latency-bound and share the chip well). fp32: all of them — every GEMM there is bound by the fp32
    matrix rate, nothing complementary runs beside it (config 4: 84.5 vs 85.9 ms/step)."""
    return bf16_fill if ctx.code == _lib.BF16 else 1.0


def _rows_per_part(B: int, H: int) -> int:
    """Image rows folded into one partial-sum row by the row reductions: keep >= ~1024 work-groups in flight
    but hand the per-image fold (one work-group per image) at most 64 rows."""
    return max(1, (B * H) // 1024, -(-H // 64))


def _wgrad_nsplit(tiles: int, ksteps: int, slab_floats: int = 0, tm: int = 128, tn: int = 128, esize: int = 2,
                  taps_per_wg: int = 1, fill: float = 1.0) -> int:
    """Split-K factor for the weight-gradient GEMM. The grid is tiles*nsplit work-groups at two per CU
    (512 slots): pick the factor that minimises an estimate of
      GEMM time / (slot quantisation efficiency * main-loop share) + slab fold traffic."""
    lds = 2 * 64 * (tm + tn) * esize + 128            # WgradCfg::LDS_BYTES
    if taps_per_wg == 3:
        lds = 2 * (72 * tm + 64 * tn) * esize          # Wgrad3Cfg::LDS_BYTES
    slots = 256 * max(1, min(4, (160 * 1024) // lds))
    if taps_per_wg == 3 and tm * tn >= 128 * 128:
        slots = 256                                    # 8-wave work-groups, one per CU
    slots = max(1, int(slots * fill))
    flops_per_step = 2.0 * tm * tn * 64 * taps_per_wg
    best, best_t = 1, float("inf")
    for n in range(1, max(1, min(ksteps // 4, 256)) + 1):
        steps = -(-ksteps // n)
        grid = tiles * n
        waves = -(-grid // slots)
        quant = grid / (waves * slots)
        t_gemm = tiles * n * steps * flops_per_step / 700e12 / (quant * steps / (steps + 3.0))
        t_fold = (n * slab_floats * 4 / 3e12) if n > 1 else 0.0
        t = t_gemm + t_fold
        if t < best_t:
            best, best_t = n, t
    return best


GROUP_ALIGN = 64        # elements: every backward stage of the flat buffers ends on a multiple of this, so that any
                        # union of stages (a DP bucket) splits into <= 16 equal, 16-byte aligned rank shards


def flat_layout(groups: List[List[torch.nn.Parameter]]):
    """Offsets of the parameters inside a flat fp32 buffer: 16-byte aligned views in the given order, each group
    (= backward stage) padded to GROUP_ALIGN elements. Returns (offsets per parameter in order, padded group sizes,
    total)."""
    offs, sizes, total = [], [], 0
    for g in groups:
        begin = total
        for p in g:
            offs.append(total)
            total += _round_up(p.numel(), 4)
        total = _round_up(total, GROUP_ALIGN)
        sizes.append(total - begin)
    return offs, sizes, total


class GradSink:
    """Flat fp32 gradient buffer: one 16-byte-aligned view per parameter, in backward order."""

    def __init__(self, ctx: Ctx, params, groups: Optional[List[List[torch.nn.Parameter]]] = None):
        if groups is None:
            groups = [list(params)]
        self.params = [p for g in groups for p in g]
        self.groups = [list(g) for g in groups]         # parameters per backward stage, in completion order
        offs, self.group_sizes, total = flat_layout(groups)
        self.flats = [torch.zeros(total, dtype=torch.float32, device=ctx.device) for _ in range(2)]
        self.views = [{id(p): f[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, offs)} for f in self.flats]
        self.active = 0

    def select(self) -> None:
        """Pick the flat buffer that no live `.grad` aliases (see UNet backward)."""
        for which in (0, 1):
            base, n = self.flats[which].data_ptr(), self.flats[which].numel() * 4
            if not any(p.grad is not None and base <= p.grad.data_ptr() < base + n for p in self.params):
                self.active = which
                return
        self.active = 0

    def view(self, p: torch.nn.Parameter) -> torch.Tensor:
        return self.views[self.active][id(p)]

    def flat(self) -> torch.Tensor:
        return self.flats[self.active]


def double_conv_params(mod) -> List[torch.nn.Parameter]:
    seq = mod.double_conv
    ps = [seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias,
          seq[3].weight, seq[3].bias, seq[4].weight, seq[4].bias]
    if len(seq) > 6:
        ps += [seq[6].fc[0].weight, seq[6].fc[2].weight]
    return ps


def grad_groups(net) -> List[List[torch.nn.Parameter]]:
    """Parameters of a UNet grouped by the backward stage that completes their gradients, in completion order
    (the layout of the flat gradient buffer, of the DP buckets and of the sharded optimizer's parameter buffer)."""
    ups = [net.up1, net.up2, net.up3, net.up4]
    convs = [net.conv1, net.conv2, net.conv3, net.conv4]
    encs = [net.inc, net.down1[1], net.down2[1], net.down3[1], net.down4[1]]
    groups = [[net.outc.weight, net.outc.bias] + double_conv_params(convs[3]) + [ups[3].weight, ups[3].bias]]
    for i in (2, 1, 0):
        groups.append(double_conv_params(convs[i]) + [ups[i].weight, ups[i].bias])
    for l in (4, 3, 2, 1, 0):
        groups.append(double_conv_params(encs[l]))
    return groups


class ConvBN:
    """conv3x3 (bias) -> BatchNorm2d -> ReLU, the unit DoubleConv is made of (:81-86)."""

    def __init__(self, ctx: Ctx, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, x: Act, name: str):
        self.ctx, self.conv, self.bn, self.x, self.name = ctx, conv, bn, x, name
        B, H, W = x.B, x.H, x.W
        self.cin, self.cout = conv.in_channels, conv.out_channels
        if x.c_len != self.cin:
            raise _lib.InsarError(f"{name}: input slice has {x.c_len} channels, conv expects {self.cin}")
        self.small = self.cin <= 4
        self._small_part = None      # slab of the first layer's weight gradient when it runs on the main stream
        if not self.small and self.cin % 64:
            raise _lib.InsarError(f"{name}: in_channels={self.cin} must be <=4 or a multiple of 64 on the HIP path")
        if self.cout % 64:
            raise _lib.InsarError(f"{name}: out_channels={self.cout} must be a multiple of 64 on the HIP path")
        self.M = B * H * W
        self.y = Act.alloc(B, H, W, self.cout, ctx.dtype, ctx.device)           # raw conv output (no bias)
        # large grids: flat-padded kernel (A rows shared by the three dx taps); else the per-tap implicit GEMM
        # 64 -> 64 channels in bf16: persistent register-weight kernel (forward and input gradient)
        c64_mode = os.environ.get("INSAR_C64", "1")           # diagnostic: 0 = off, fwd / bwd = that direction only
        self.c64 = (not self.small) and bool(call("insar_conv3x3_c64_ok", x.ref, self.cout)) and c64_mode != "0"
        self.c64_fwd, self.c64_bwd = self.c64 and c64_mode != "bwd", self.c64 and c64_mode != "fwd"
        self.flat_fwd = (not self.small) and not self.c64 and bool(call("insar_conv3x3_flat_ok", x.ref, self.cout))
        self.flat_bwd = (not self.small) and not self.c64 and bool(call("insar_conv3x3_flat_ok", x.ref, self.cin))
        # the rest: per-tap implicit GEMM, or (bf16, where that kernel would not run its 256 x 256 tiles) the flat kernel's row tiles
        plain = not (self.small or self.c64)
        self.rows_fwd = _rows_flags(x, self.cout) if (plain and not self.flat_fwd) else 0
        self.rows_bwd = _rows_flags(x, self.cin, self.cout, flip=1) if (plain and not self.flat_bwd) else 0
        if self.small:
            self.stat_rows = call("insar_conv3x3_small_fwd_rows", x.ref, self.y.ref)
        elif self.rows_fwd:
            self.stat_rows = call("insar_conv3x3_flat_stat_rows", x.ref, self.cout, self.rows_fwd)
        elif self.c64_fwd:
            self.stat_rows = call("insar_conv3x3_c64_rows", x.ref)
        elif self.c64:
            self.stat_rows = call("insar_igemm_num_mtiles", self.M, self.cout)
        elif self.flat_fwd:
            self.stat_rows = call("insar_conv3x3_flat_stat_rows", x.ref, self.cout, _flat_flags(0, x))
        else:
            self.stat_rows = call("insar_igemm_num_mtiles", self.M, self.cout)
        self.stats = ctx.f32(self.stat_rows, 2, self.cout)
        # BN partial sums: slabs with many rows are folded to <= 64 rows first, bn_finalize folds the rest
        self.stat_rps = 0 if self.stat_rows <= STAT_PREFOLD_ROWS else max(64, -(-self.stat_rows // 64))
        self.fold_rows = self.stat_rows if not self.stat_rps else -(-self.stat_rows // self.stat_rps)
        self.sums = ctx.f32(self.fold_rows, 2, self.cout) if self.stat_rps else self.stats
        self.scale, self.shift = ctx.f32(self.cout), ctx.f32(self.cout)
        self.mean, self.invstd = ctx.f32(self.cout), ctx.f32(self.cout)
        self.k12 = ctx.f32(2, self.cout)                  # one buffer: SyncBN all-reduces both coefficient vectors at once
        self.k1, self.k2 = self.k12[0], self.k12[1]
        self._ticket = None                               # last-arriver counter of the single-launch coefficient stages
        self.sync_sums = None                             # [2][C] batch sums of this replica (SyncBN only)
        self.sync = None
        self.red_rpp = _rows_per_part(B, H)
        self.red_rows = -(-H // self.red_rpp)
        self.red_part = ctx.f32(B * self.red_rows, 2, self.cout)
        self.bwd_ws = ctx.f32(B * (3 * self.cout + max(self.cout // 16, 1)))
        self.dy = None            # gradient wrt the raw conv output (allocated on first backward)
        self.w = None if self.small else GemmWeight(ctx, conv.weight, "conv3")
        # BatchNorm-backward sums written by the epilogue of the GEMM that produces this unit's incoming gradient
        # (InsarBstat): slab [B * bred_rows][2][C]; bred_ready is set by that GEMM's launch and consumed by backward()
        self.bred, self.bred_rows, self.bred_ready = None, 0, False

    def bstat_slab(self, rows_total: int, per_image: bool):
        """(slab, (y, scale, shift)) for a producer GEMM whose statistics slab has `rows_total` rows, or None if this unit
        cannot take its backward sums from it. per_image: the rows are whole-image groups in image order (an SE unit's
        coefficient stage folds them per image); else any partition will do and the slab is padded with zero rows to a
        multiple of B (the coefficient stage sums B equal groups)."""
        B = self.x.B
        if per_image and rows_total % B:
            return None
        rows = -(-rows_total // B)
        if self.bred is None or self.bred_rows != rows:
            self.bred, self.bred_rows = self.ctx.f32(B * rows, 2, self.cout), rows
        return self.bred, (self.y, self.scale, self.shift)

    # ---- forward: y = conv(x); BN statistics; scale/shift ---------------------------------------
    def forward_conv(self, training: bool, sync=None) -> None:
        """sync = (process group, world size): synchronised BatchNorm — the batch statistics are those of the GLOBAL batch
        (one all-reduce of this layer's [sum, sum of squares] per forward, one of [k1, k2] per backward), which makes
        data-parallel training equal to the reference's single-device batch ('BatchNorm under DP')."""
        s = _lib.stream_ptr()
        self.sync = sync if training else None
        if training and self.M <= 1 and not sync:
            # nn.BatchNorm2d's own check (torch.nn.functional._verify_batch_size), same exception and text
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{self.x.B}, {self.cout}, {self.x.H}, {self.x.W}])")
        if self.small:
            w = self.conv.weight.detach()
            call("insar_conv3x3_small_fwd", self.x.ref, ptr(w), self.y.ref, ptr(self.stats) if training else 0, s)
        elif self.c64_fwd:
            self.ctx.join_side()          # GEMM-layout weights come from the side stream (UNetPlan.forward)
            _conv3x3_c64(self.x, self.y, self.w.fwd(), 0, self.stats if training else None)
        elif self.c64:
            self.ctx.join_side()
            _igemm(self.x, self.y, self.w.fwd(), self.cout, self.x.H, self.x.W, 1, _TAPS3, 0,
                   stats=self.stats if training else None)
        elif self.flat_fwd:
            self.ctx.join_side()
            _conv3x3_flat(self.x, self.y, self.w.fwd(), 0, self.stats if training else None)
        elif self.rows_fwd:
            self.ctx.join_side()
            _conv3x3_flat(self.x, self.y, self.w.fwd(), 0, self.stats if training else None, geo=self.rows_fwd)
        else:
            self.ctx.join_side()
            _igemm(self.x, self.y, self.w.fwd(), self.cout, self.x.H, self.x.W, 1, _TAPS3, 0,
                   stats=self.stats if training else None)
        if training and self.stat_rps:
            call("insar_colsum_partial", ptr(self.stats), ptr(self.sums), self.stat_rows, 2 * self.cout, self.stat_rps, s)
        bn = self.bn
        d = InsarBnFinalize()
        d.part, d.rows, d.count, d.C, d.training = ptr(self.sums), self.fold_rows, self.M, self.cout, int(training)
        if self.sync:
            import torch.distributed as dist
            pg, world = self.sync
            if self.sync_sums is None:
                self.sync_sums = self.ctx.f32(2 * self.cout)
            self.ctx.colsum(self.sums, self.sync_sums, 1, self.fold_rows, 2 * self.cout)
            dist.all_reduce(self.sync_sums, op=dist.ReduceOp.SUM, group=pg)
            d.part, d.rows, d.count = ptr(self.sync_sums), 1, self.M * world
        d.conv_bias = ptr(self.conv.bias) if self.conv.bias is not None else 0
        d.gamma, d.beta = ptr(bn.weight), ptr(bn.bias)
        d.running_mean, d.running_var = ptr(bn.running_mean), ptr(bn.running_var)
        d.num_batches_tracked = ptr(bn.num_batches_tracked)
        d.momentum = bn.momentum if bn.momentum is not None else 0.1
        d.eps = bn.eps
        d.scale, d.shift, d.mean, d.invstd = ptr(self.scale), ptr(self.shift), ptr(self.mean), ptr(self.invstd)
        call("insar_bn_finalize", C.byref(d), s)

    def apply(self, dst: Act, gate: Optional[torch.Tensor], pooled: Optional[Act] = None,
              pool_arg: Optional[torch.Tensor] = None) -> None:
        if pooled is not None and ((self.y.H | self.y.W) & 1):
            # odd grid (tile size not a multiple of 16): nn.MaxPool2d(2) drops the last row / column; two passes
            call("insar_bn_relu_apply", self.y.ref, ptr(self.scale), ptr(self.shift), ptr(gate), dst.ref, 1, _lib.stream_ptr())
            call("insar_maxpool2_fwd", dst.ref, pooled.ref, _lib.stream_ptr())
            return
        if pooled is not None:          # encoder block: the 2x2 max-pool of dst comes out of the same pass
            if pool_arg is not None:    # ... with its arg-max map, for the backward passes of this unit
                call("insar_bn_relu_apply_pool_arg", self.y.ref, ptr(self.scale), ptr(self.shift), ptr(gate), dst.ref,
                     pooled.ref, ptr(pool_arg), 1, _lib.stream_ptr())
            else:
                call("insar_bn_relu_apply_pool", self.y.ref, ptr(self.scale), ptr(self.shift), ptr(gate), dst.ref,
                     pooled.ref, 1, _lib.stream_ptr())
            return
        call("insar_bn_relu_apply", self.y.ref, ptr(self.scale), ptr(self.shift), ptr(gate), dst.ref, 1,
             _lib.stream_ptr())

    def _coef(self, coef_args, s, se) -> None:
        """The backward coefficients. A unit without an SE gate: one channel-parallel launch over all slab rows
        (insar_bn_bwd_coef; COEF_SIMPLE=0: the per-image stage + batch fold in one launch, stage 2 by the work-group that
        finishes stage 1 last). With a gate: two launches (stage 2 carries the SE weight gradients)."""
        if COEF_SIMPLE and se is None:       # no SE gate: the coefficients are linear in the slab rows — one channel-parallel launch
            call("insar_bn_bwd_coef", coef_args[0], coef_args[1], coef_args[2] * self.x.B, coef_args[3], coef_args[6], coef_args[7], s)
        elif COEF_FUSE and se is None:
            if self._ticket is None:
                self._ticket = torch.zeros(1, dtype=torch.int32, device=self.ctx.device)
            call("insar_bnse_bwd_coef_fused", *coef_args, ptr(self._ticket), s)
        else:
            call("insar_bnse_bwd_coef", *coef_args, s)

    def _sync_k(self) -> None:
        """SyncBN backward: k1 = mean(g*mask), k2 = mean(g*mask*xhat) over the GLOBAL batch = the mean over the replicas of
        their local means (equal local batch sizes)."""
        if getattr(self, "sync", None):
            import torch.distributed as dist
            pg, world = self.sync
            if dist.get_backend(pg) == "nccl":
                dist.all_reduce(self.k12, op=dist.ReduceOp.AVG, group=pg)
            else:
                dist.all_reduce(self.k12, op=dist.ReduceOp.SUM, group=pg)
                self.k12.div_(world)

    # ---- backward -------------------------------------------------------------------------------
    def backward(self, dout: Optional[Act], sink: GradSink, training: bool, se: Optional["SEState"], dx: Optional[Act],
                 outc_grad=None, pool_grad=None, bstat_for: Optional["ConvBN"] = None, bstat_se: bool = False) -> None:
        """dout: gradient wrt this unit's output (after ReLU and, if `se`, the SE gate); or `outc_grad` =
        (dlogits, outc weight, K) when this unit feeds the 1x1 output conv: the reduce and apply passes then recompute
        that gradient from dlogits instead of reading a 64-channel tensor (csrc/pointwise.hip, OutcGrad)."""
        ctx, s = self.ctx, _lib.stream_ptr()
        B, H, W = self.x.B, self.x.H, self.x.W
        # the network's first layer: nobody but its own weight gradient reads dy, so the apply pass is evaluated inside the
        # weight-gradient kernel and dy is never written (insar_conv3x3_small_wgrad_fused; bit for bit the two launches)
        fuse_small = (self.small and SMALL_WGRAD_FUSE and SMALL_WGRAD_MAIN and dx is None and se is None and outc_grad is None
                      and pool_grad is None and dout is not None and not (training and SPLIT_COEF)
                      and call("insar_conv3x3_small_wgrad_fused_ok", self.x.ref, self.y.ref) == 1
                      and call("insar_conv3x3_small_wgrad_fused_ok", self.x.ref, dout.ref) == 1)
        if self.dy is None and not fuse_small:
            self.dy = Act.alloc(B, H, W, self.cout, ctx.dtype, ctx.device)
        if outc_grad is not None:
            # (dlogits, outc weight, K[, gate, wpart]): with wpart the reduce pass also writes the output conv's own
            # parameter-gradient partials, one row per row part (OutConvPlan.backward_fused folds them)
            dl, wout, K = outc_grad[:3]
            og_gate, og_wpart = (outc_grad[3], outc_grad[4]) if len(outc_grad) > 3 else (None, None)
            call("insar_bnrelu_bwd_reduce_outc", ptr(dl), ptr(wout), K, self.y.ref, ptr(self.scale), ptr(self.shift),
                 ptr(self.red_part), 1, self.red_rpp, ptr(og_gate), ptr(og_wpart), s)
        elif pool_grad is not None:
            # encoder block: dout = skip gradient + the max-pool gradient routed by the forward arg-max map, summed on the fly
            dpool, parg = pool_grad
            call("insar_bnrelu_bwd_reduce_pool", dout.ref, dpool.ref, ptr(parg), self.y.ref, ptr(self.scale), ptr(self.shift),
                 ptr(self.red_part), 1, self.red_rpp, s)
        elif self.bred_ready:
            pass                  # the sums came out of the epilogue of the GEMM that wrote dout (bstat_slab)
        else:
            call("insar_bnrelu_bwd_reduce", dout.ref, self.y.ref, ptr(self.scale), ptr(self.shift), ptr(self.red_part), 1, self.red_rpp, s)
        red, red_rows = (self.bred, self.bred_rows) if (self.bred_ready and outc_grad is None and pool_grad is None) else (self.red_part, self.red_rows)
        self.bred_ready = False
        d = InsarBnSeBwd()
        d.B, d.H, d.W, d.C = B, H, W, self.cout
        d.Cr = se.cr if se else 1
        d.use_se = 1 if se else 0
        d.mean, d.invstd = ptr(self.mean), ptr(self.invstd)
        if se:
            d.pooled, d.sq, d.hid, d.gate = ptr(se.pooled), ptr(se.sq), ptr(se.hid), ptr(se.gate)
            d.w1, d.w2 = ptr(se.fc1.weight), ptr(se.fc2.weight)
            d.dw1, d.dw2 = ptr(sink.view(se.fc1.weight)), ptr(sink.view(se.fc2.weight))
            d.coefB = ptr(se.coefB)
        d.dgamma, d.dbeta = ptr(sink.view(self.bn.weight)), ptr(sink.view(self.bn.bias))
        d.k1, d.k2 = ptr(self.k1), ptr(self.k2)
        d.accumulate = 0
        dbias = ptr(sink.view(self.conv.bias)) if self.conv.bias is not None else 0
        coef_args = (C.byref(d), ptr(red), red_rows, ptr(self.scale), ptr(self.shift), ptr(self.bwd_ws),
                     dbias, int(training))
        # Training mode: only the per-image stage stays on the dgrad chain; the apply pass folds k1 / k2 from its
        # partial sums itself and the batch fold (parameter gradients) follows the weight gradient on the side stream.
        split = training and SPLIT_COEF and self.cout <= 1024 and outc_grad is None and pool_grad is None and not self.sync
        stage2 = None
        if split:
            call("insar_bnse_bwd_coef_stage", *coef_args, 1, s)
            tb = self.bwd_ws.data_ptr() + 4 * B * (self.cout + d.Cr)
            call("insar_bnrelu_bwd_apply_part", dout.ref, self.y.ref, ptr(self.scale), ptr(self.shift), ptr(self.mean),
                 ptr(self.invstd), ptr(se.gate) if se else 0, ptr(se.coefB) if se else 0, tb, tb + 4 * B * self.cout,
                 self.dy.ref, 1, s)
            stage2 = lambda: call("insar_bnse_bwd_coef_stage", *coef_args, 2, _lib.stream_ptr())
        elif outc_grad is not None:
            self._coef(coef_args, s, se)
            self._sync_k()
            call("insar_bnrelu_bwd_apply_outc", ptr(dl), ptr(wout), K, self.y.ref, ptr(self.scale), ptr(self.shift),
                 ptr(self.mean), ptr(self.invstd), ptr(se.gate) if se else 0, ptr(se.coefB) if se else 0, ptr(self.k1),
                 ptr(self.k2), self.dy.ref, 1, s)
        elif pool_grad is not None:
            self._coef(coef_args, s, se)
            self._sync_k()
            call("insar_bnrelu_bwd_apply_pool", dout.ref, dpool.ref, ptr(parg), self.y.ref, ptr(self.scale), ptr(self.shift),
                 ptr(self.mean), ptr(self.invstd), ptr(se.gate) if se else 0, ptr(se.coefB) if se else 0, ptr(self.k1),
                 ptr(self.k2), self.dy.ref, 1, s)
        else:
            self._coef(coef_args, s, se)
            self._sync_k()
            if not fuse_small:
                call("insar_bnrelu_bwd_apply", dout.ref, self.y.ref, ptr(self.scale), ptr(self.shift), ptr(self.mean),
                     ptr(self.invstd), ptr(se.gate) if se else 0, ptr(se.coefB) if se else 0, ptr(self.k1), ptr(self.k2),
                     self.dy.ref, 1, s)
        # weight gradient (side stream: reads x and dy, writes only the gradient sink)
        gw = sink.view(self.conv.weight)

        def small_weight_grad(part):
            nb = call("insar_conv3x3_small_wgrad_blocks", B, H)
            cols = self.cout * self.cin * 9
            if part is None:
                part = ctx.wgrad_part(nb * cols)
            elif part.numel() < nb * cols:
                part = self._small_part = ctx.f32(nb * cols)
            if fuse_small:
                call("insar_conv3x3_small_wgrad_fused", self.x.ref, dout.ref, self.y.ref, ptr(self.scale), ptr(self.shift),
                     ptr(self.mean), ptr(self.invstd), ptr(self.k1), ptr(self.k2), 1, ptr(part), _lib.stream_ptr())
            else:
                call("insar_conv3x3_small_wgrad", self.x.ref, self.dy.ref, ptr(part), _lib.stream_ptr())
            ctx.colsum(part, gw, 1, nb, cols)

        def weight_grad():
            if self.small and SMALL_WGRAD_MAIN and stage2 is None and dx is None:
                # the first layer of the network is the LAST unit of backward: nothing is left on the main stream for its weight
                # gradient to run beside, and on the side stream it costs two cross-queue hand-offs (about 12 us each) at the
                # very end of the step; it runs on the main stream, with a slab of its own (the side stream's is still in use)
                if self._small_part is None:
                    self._small_part = ctx.f32(1)
                small_weight_grad(self._small_part)
                return
            with ctx.side_stream():
                if self.small:
                    small_weight_grad(None)
                else:
                    _wgrad_conv3(ctx, self.x, self.dy, gw)
                if stage2 is not None:
                    stage2()

        # Issue order: with WGRAD_LATE the side stream picks the weight gradient up only once the input-gradient
        # GEMM of this unit is enqueued, so that it runs beside the NEXT unit's HBM- / latency-bound coefficient
        # chain instead of competing with the dgrad (both MFMA-bound) for the CUs.
        if not WGRAD_LATE:
            weight_grad()
        if dx is not None:
            if self.small:
                raise _lib.InsarError(f"{self.name}: input gradient of the direct first-layer conv is not provided")
            if self.c64_bwd:
                slab = None
                if bstat_for is not None and BSTAT_FUSE and BSTAT_C64 and not bstat_se and _same_layout(dx, bstat_for.y):
                    slab = bstat_for.bstat_slab(call("insar_conv3x3_c64_rows", self.dy.ref), False)
                if slab:
                    _conv3x3_c64(self.dy, dx, self.w.dgrad(), 1, slab[0], bstat=slab[1])
                    bstat_for.bred_ready = True
                else:
                    _conv3x3_c64(self.dy, dx, self.w.dgrad(), 1, None)
            elif self.flat_bwd and not self.c64:
                slab = None
                if bstat_for is not None and BSTAT_FUSE and not bstat_se and _same_layout(dx, bstat_for.y):
                    slab = bstat_for.bstat_slab(call("insar_conv3x3_flat_stat_rows", self.dy.ref, self.cin, _flat_flags(1, self.dy)), False)
                if slab:
                    _conv3x3_flat(self.dy, dx, self.w.dgrad(), 1, slab[0], bstat=slab[1])
                    bstat_for.bred_ready = True
                else:
                    _conv3x3_flat(self.dy, dx, self.w.dgrad(), 1, None)
            elif self.rows_bwd:
                # row tiles are whole image rows of ONE image: the slab rows group per image, as an SE consumer needs
                slab = None
                geo = self.rows_bwd & ~4 if bstat_se else self.rows_bwd      # (persistent work-groups' carried sums mix images)
                if bstat_for is not None and BSTAT_FUSE and _same_layout(dx, bstat_for.y):
                    slab = bstat_for.bstat_slab(call("insar_conv3x3_flat_stat_rows", self.dy.ref, self.cin, geo), bstat_se)
                _conv3x3_flat(self.dy, dx, self.w.dgrad(), 1, slab[0] if slab else None, bstat=slab[1] if slab else None, geo=geo)
                if slab:
                    bstat_for.bred_ready = True
            else:
                slab = _igemm_bstat_slab(bstat_for, bstat_se, self.M, self.cin, H * W, dx)
                _igemm(self.dy, dx, self.w.dgrad(), self.cin, H, W, 1, _TAPS3_DGRAD, 0,
                       stats=slab[0] if slab else None, bstat=slab[1] if slab else None)
                if slab:
                    bstat_for.bred_ready = True
        if WGRAD_LATE:
            weight_grad()


def _same_layout(a: Act, b: Act) -> bool:
    return (a.buf.shape == b.buf.shape and a.buf.dtype == b.buf.dtype and a.C == b.C and a.c_off == b.c_off
            and a.c_len == b.c_len)


def _igemm_bstat_slab(consumer: Optional["ConvBN"], se: bool, M: int, N: int, hw: int, out: Act):
    """Statistics slab + InsarBstat operands for an insar_igemm launch (mode 0, dense output `out`, M rows, N columns) whose
    output is `consumer`'s incoming gradient, or None: BSTAT_FUSE off, layouts differ, or — for a unit with an SE gate,
    whose coefficient stage needs the sums per image — GEMM row tiles that straddle images."""
    if consumer is None or not BSTAT_FUSE or not _same_layout(out, consumer.y):
        return None
    bm = call("insar_igemm_tile_rows", M, N)
    if se and hw % bm:
        return None
    return consumer.bstat_slab(call("insar_igemm_num_mtiles", M, N), se)


def _wgrad_conv3(ctx: Ctx, x: Act, dy: Act, grad: torch.Tensor) -> None:
    B, H, W = x.B, x.H, x.W
    cin, cout = x.c_len, dy.c_len
    pair = call("insar_wgrad_conv3_tile", x.ref, cout) if WGRAD_ROWS else 0
    pairx = call("insar_wgrad_conv3x_tile", x.ref, cout) if (WGRAD_ROWS and WGRAD_X) else 0
    pairy = call("insar_wgrad_conv3y_tile", x.ref, cout) if (WGRAD_ROWS and ((WGRAD_Y & 1 and pairx) or (WGRAD_Y & 2 and not pairx))) else 0
    pairk = call("insar_wgrad_conv3k_tile", x.ref, cout) if (WGRAD_ROWS and WGRAD_K and not pairx) else 0
    if pairk and "%dx%d" % (pairk >> 16, pairk & 0xffff) not in WGRAD_K_TILES:
        pairk = 0
    if pairk:
        # 64 / 128 channels a side (csrc/wgrad3k.hip): one round of work-groups over the share of the chip the launch aims at;
        # a work-group writes KS slabs (its waves split the pixels of a K step)
        tm, tn = pairk >> 16, pairk & 0xffff
        ks = call("insar_wgrad_conv3k_slices", x.ref, cout)
        tiles = 3 * (cin // tm) * (cout // tn)
        fill = _side_fill(ctx, WGRAD_FILL) if (ctx.side is not None and not (PROFILER is not None and PROFILER.alone)) else WGRAD_FILL_ALONE
        ksteps = B * H * W // (ks * 32)
        nsplit = max(1, min(int(256 * fill) // tiles, ksteps))
        part = ctx.wgrad_part(nsplit * ks * 9 * cout * cin)
        if PROFILER is not None:
            nbytes = ctx.esize * B * H * W * (cin + cout) + 4.0 * nsplit * ks * 9 * cout * cin
            PROFILER.run("wgrad3k_kernel<%d, %d>" % (tm, tn), 2.0 * B * H * W * cin * cout * 9,
                         lambda: call("insar_wgrad_conv3k", x.ref, dy.ref, ptr(part), nsplit, _lib.stream_ptr()), nbytes)
        else:
            call("insar_wgrad_conv3k", x.ref, dy.ref, ptr(part), nsplit, _lib.stream_ptr())
        ctx.wgrad_finish(part, grad, nsplit * ks, 9, cout, cin, 0)
        return
    if pairy:
        # 128 x 128 tiles, two 4-wave work-groups per CU: the launch aims at twice the work-group count of the one-per-CU kernels
        tiles = 3 * (cin // 128) * (cout // 128)
        fill = _side_fill(ctx, WGRAD_FILL) if (ctx.side is not None and not (PROFILER is not None and PROFILER.alone)) else WGRAD_FILL_ALONE
        ksteps = B * H * W // WG_BKP
        nsplit = max(1, min(int(512 * fill) // tiles, ksteps // 4))
        part = ctx.wgrad_part(nsplit * 9 * cout * cin)
        if PROFILER is not None:
            nbytes = ctx.esize * B * H * W * (cin + cout) + 4.0 * nsplit * 9 * cout * cin
            PROFILER.run("wgrad3y_kernel<128, 128>", 2.0 * B * H * W * cin * cout * 9,
                         lambda: call("insar_wgrad_conv3y", x.ref, dy.ref, ptr(part), nsplit, _lib.stream_ptr()), nbytes)
        else:
            call("insar_wgrad_conv3y", x.ref, dy.ref, ptr(part), nsplit, _lib.stream_ptr())
        ctx.wgrad_finish(part, grad, nsplit, 9, cout, cin, 0)
        return
    if pair or pairx:
        # three taps of a kernel row per work-group (csrc/wgrad3.hip): a third of the operand staging; where one side has
        # 256 channels and the other 128, the 256 x 128 tile kernel with the six-phase K loop (csrc/wgrad3x.hip)
        entry = "insar_wgrad_conv3x" if pairx else "insar_wgrad_conv3"
        pair = pairx or pair
        tm, tn = pair >> 16, pair & 0xffff
        tiles = 3 * (cin // tm) * (cout // tn)
        # Beside the dgrad chain (side stream) the weight gradient should fill about HALF the work-group slots: the
        # main stream's kernels keep CUs, there are half as many slabs to fold, and the launch still ends before the
        # next one is due (same-box sweep of the fill factor: 1.0 8.09-8.14, 0.7 7.86, 0.5 7.81-7.86, 0.35 7.86, 0.25
        # 9.26 ms/step). Alone on the GPU (single-stream runs, the per-kernel event pass of bench.py) it fills the chip.
        fill = _side_fill(ctx, WGRAD_FILL if pairx else WGRAD_FILL_SMALL) if (ctx.side is not None and not (PROFILER is not None and PROFILER.alone)) else WGRAD_FILL_ALONE
        nsplit = _wgrad_nsplit(tiles, B * H * W // WG_BKP, 9 * cout * cin, tm, tn, ctx.esize, taps_per_wg=3, fill=fill)
        if WGRAD_GRID_CAP and fill < 1.0 and tm * tn >= 128 * 128:
            # deep layers (48 - 192 tiles): the cost model lands on 240 - 384 one-per-CU work-groups; beside the dgrad chain a
            # grid that leaves CUs to the main stream does better (same-box sweep, tiles/s: no cap 2 110, <= 256: 2 117,
            # <= 200: 2 126, <= 160: 2 082, <= 128: 2 054)
            nsplit = min(nsplit, max(1, WGRAD_GRID_CAP // tiles))
        part = ctx.wgrad_part(nsplit * 9 * cout * cin)
        if PROFILER is not None:
            if pairx:
                tag = "wgrad3x_kernel<%d, %d>" % (tm, tn)
            else:
                tag = "wgrad3_kernel<%s, %d, %d, %d>" % ("float" if ctx.code == _lib.F32 else "bf16_t", tm, tn,
                                                         8 if tm == 128 and tn == 128 else 4)
            # algorithmic bytes: both operands read once, the split-K slabs written once
            nbytes = ctx.esize * B * H * W * (cin + cout) + 4.0 * nsplit * 9 * cout * cin
            PROFILER.run(tag, 2.0 * B * H * W * cin * cout * 9,
                         lambda: call(entry, x.ref, dy.ref, ptr(part), nsplit, _lib.stream_ptr()), nbytes)
        else:
            call(entry, x.ref, dy.ref, ptr(part), nsplit, _lib.stream_ptr())
        ctx.wgrad_finish(part, grad, nsplit, 9, cout, cin, 0)
        return
    tabx = ctx.pixel_table(B, H, W, 1, H, W, W + 3)      # taps move on x: tail = first interior pixel
    tabdy = ctx.pixel_table(B, H, W, 1, H, W, 0)         # tail = zero halo pixel
    mpad = tabx.numel()
    tm, tn = _wgrad_tiles(cin, cout, ctx.code)
    tiles = 9 * (cin // tm) * (cout // tn)
    nsplit = _wgrad_nsplit(tiles, mpad // WG_BKP, 9 * cout * cin, tm, tn, ctx.esize)
    part = ctx.wgrad_part(nsplit * 9 * cout * cin)
    d = InsarWgrad()
    d.x, d.dy = x.desc, dy.desc
    d.tabx, d.tabdy, d.part = ptr(tabx), ptr(tabdy), ptr(part)
    d.Mpad, d.nsplit, d.ntaps = mpad, nsplit, 9
    for i, (ty, tx) in enumerate(_TAPS3):
        d.offx[i] = ty * (W + 2) + tx
        d.offdy[i] = 0
    _launch_wgrad(d, B * H * W, cin, cout, 9, ctx.code)
    ctx.wgrad_finish(part, grad, nsplit, 9, cout, cin, 0)


class SEState:
    """Buffers of one SELayer (:45-72) attached to the second ConvBN of a DoubleConv."""

    def __init__(self, ctx: Ctx, se_module, B: int, H: int, Cn: int):
        self.fc1, self.fc2 = se_module.fc[0], se_module.fc[2]
        self.cr = self.fc1.out_features
        self.rpp = _rows_per_part(B, H)
        self.rows = -(-H // self.rpp)
        self.part = ctx.f32(B * self.rows, 2, Cn)
        self.pooled = ctx.f32(B, 2, Cn)
        self.sq, self.gate, self.coefB = ctx.f32(B, Cn), ctx.f32(B, Cn), ctx.f32(B, Cn)
        self.hid = ctx.f32(B, self.cr)


class DoubleConvPlan:
    """[conv3x3 -> BN -> ReLU] x 2 (+ SELayer) = DoubleConv.forward (:75-97)."""

    def __init__(self, ctx: Ctx, mod, x: Act, out: Act, name: str):
        self.ctx, self.mod, self.x, self.out, self.name = ctx, mod, x, out, name
        self.pool_out: Optional[Act] = None      # set by the plan for encoder blocks: MaxPool2d(2) of `out`
        self.pool_arg: Optional[torch.Tensor] = None   # ... and its arg-max byte map (B, H/2, W/2, C), see POOL_FUSE
        seq = mod.double_conv
        self.u1 = ConvBN(ctx, seq[0], seq[1], x, name + ".0")
        self.z1 = Act.alloc(x.B, x.H, x.W, self.u1.cout, ctx.dtype, ctx.device)
        self.u2 = ConvBN(ctx, seq[3], seq[4], self.z1, name + ".3")
        self.se = SEState(ctx, seq[6], x.B, x.H, self.u2.cout) if len(seq) > 6 else None
        if out.c_len != self.u2.cout:
            raise _lib.InsarError(f"{name}: output slice has {out.c_len} channels, expected {self.u2.cout}")
        self.dz1 = None

    def params(self) -> List[torch.nn.Parameter]:
        return double_conv_params(self.mod)

    def forward(self, training: bool, outc: Optional["OutConvPlan"] = None, sync=None) -> Optional[torch.Tensor]:
        """outc: the 1x1 output conv when this is the last block and its output goes nowhere else — the final
        BN/ReLU/gate pass then writes the logits instead of `out` (returned)."""
        s = _lib.stream_ptr()
        self.u1.forward_conv(training, sync)
        self.u1.apply(self.z1, None)
        self.u2.forward_conv(training, sync)
        if self.se:
            se, u2 = self.se, self.u2
            call("insar_se_squeeze", u2.y.ref, ptr(u2.scale), ptr(u2.shift), ptr(se.part), 1, se.rpp, s)
            d = InsarSeFwd()
            d.part, d.rows, d.pooled = ptr(se.part), se.rows, ptr(se.pooled)
            d.B, d.H, d.W, d.C, d.Cr = self.x.B, self.x.H, self.x.W, u2.cout, se.cr
            d.scale, d.shift = ptr(u2.scale), ptr(u2.shift)
            d.w1, d.w2 = ptr(se.fc1.weight), ptr(se.fc2.weight)
            d.sq, d.hid, d.gate = ptr(se.sq), ptr(se.hid), ptr(se.gate)
            call("insar_se_excite", C.byref(d), s)
            if outc is not None:
                return outc.forward_fused(u2, se.gate)
            u2.apply(self.out, se.gate, self.pool_out, self.pool_arg)
        else:
            if outc is not None:
                return outc.forward_fused(self.u2, None)
            self.u2.apply(self.out, None, self.pool_out, self.pool_arg)
        return None

    def backward(self, dout: Optional[Act], sink: GradSink, training: bool, dx: Optional[Act], outc_grad=None,
                 pool_grad=None) -> None:
        if self.dz1 is None:
            self.dz1 = Act.alloc(self.x.B, self.x.H, self.x.W, self.u1.cout, self.ctx.dtype, self.ctx.device)
        self.u2.backward(dout, sink, training, self.se, self.dz1, outc_grad, pool_grad, bstat_for=self.u1)
        self.u1.backward(self.dz1, sink, training, None, dx)


class UpPlan:
    """ConvTranspose2d(k=2, s=2) (:112,115,118,121) writing the upper half of a concat buffer."""

    def __init__(self, ctx: Ctx, mod: torch.nn.ConvTranspose2d, x: Act, out: Act, name: str):
        self.ctx, self.mod, self.x, self.out, self.name = ctx, mod, x, out, name
        self.cin, self.cout = mod.in_channels, mod.out_channels
        if self.cin % 64 or self.cout % 64:
            raise _lib.InsarError(f"{name}: channels must be multiples of 64 on the HIP path")
        self.w = GemmWeight(ctx, mod.weight, "convT")
        # The transposed conv produces (2h, 2w); the skip it is concatenated with may be one pixel larger (tile sizes that
        # are not multiples of 16): the reference then resizes bilinearly to the skip's size (:138-139 ...). `conv_out` is
        # where the GEMM writes: the concat slice itself, or a (2h, 2w) buffer that is resized into it.
        self.resize = (out.H, out.W) != (2 * x.H, 2 * x.W)
        if self.resize and not (2 * x.H <= out.H <= 2 * x.H + 1 and 2 * x.W <= out.W <= 2 * x.W + 1):
            raise _lib.InsarError(f"{name}: output {out.H}x{out.W} is not the skip size of a {x.H}x{x.W} input")
        self.conv_out = Act.alloc(x.B, 2 * x.H, 2 * x.W, self.cout, ctx.dtype, ctx.device) if self.resize else out
        self.dconv_out = None
        out_h = self.conv_out.H
        self.bias_rpp = _rows_per_part(out.B, out_h)
        self.bias_rows = out.B * -(-out_h // self.bias_rpp)
        self.bias_part = ctx.f32(self.bias_rows, 2, self.cout)
        self.bias_sum = ctx.f32(2, self.cout)

    def params(self):
        return [self.mod.weight, self.mod.bias]

    def forward(self) -> None:
        self.ctx.join_side()
        _igemm(self.x, self.conv_out, self.w.fwd(), 4 * self.cout, self.x.H, self.x.W, 1, [(0, 0)], 1,
               bias=self.mod.bias.detach() if self.mod.bias is not None else None)
        if self.resize:
            call("insar_resize_bilinear_fwd", self.conv_out.ref, self.out.ref, _lib.stream_ptr())

    def backward(self, dout: Act, sink: GradSink, dx: Optional[Act], consumer: Optional["DoubleConvPlan"] = None) -> None:
        """dout: gradient slice wrt this layer's output (upper half of the dcat buffer). consumer: the block whose output
        this layer's input is — its last unit takes its BatchNorm-backward sums from the epilogue of the input-gradient GEMM."""
        ctx, s = self.ctx, _lib.stream_ptr()
        x, B, h, w = self.x, self.x.B, self.x.H, self.x.W
        if self.resize:              # gradient of the (2h, 2w) conv output = adjoint of the bilinear resize
            if self.dconv_out is None:
                co = self.conv_out
                self.dconv_out = Act.alloc(co.B, co.H, co.W, co.c_len, ctx.dtype, ctx.device)
            call("insar_resize_bilinear_bwd", dout.ref, self.dconv_out.ref, s)
            dout = self.dconv_out
        tabx = ctx.pixel_table(B, h, w, 1, h, w, 0)
        tabdy = ctx.pixel_table(B, h, w, 2, dout.H, dout.W, 0)
        mpad = tabx.numel()
        tm, tn = _wgrad_tiles(self.cin, self.cout, ctx.code)
        tiles = 4 * (self.cin // tm) * (self.cout // tn)
        fill = _side_fill(ctx, WGRAD_FILL_T) if (ctx.side is not None and not (PROFILER is not None and PROFILER.alone)) else 1.0
        nsplit = _wgrad_nsplit(tiles, mpad // WG_BKP, 4 * self.cout * self.cin, tm, tn, ctx.esize, fill=fill)
        def weight_grad():
            with ctx.side_stream():
                if self.mod.bias is not None:     # dbias = sum over pixels of dout (mask-free row reduction)
                    call("insar_bnrelu_bwd_reduce", dout.ref, dout.ref, ptr(ctx.const(0.0, self.cout)),
                         ptr(ctx.const(1.0, self.cout)), ptr(self.bias_part), 0, self.bias_rpp, _lib.stream_ptr())
                    ctx.colsum(self.bias_part, sink.view(self.mod.bias), 1, self.bias_rows, self.cout, ld=2 * self.cout)
                part = ctx.wgrad_part(nsplit * 4 * self.cout * self.cin)
                d = InsarWgrad()
                d.x, d.dy = x.desc, dout.desc
                d.tabx, d.tabdy, d.part = ptr(tabx), ptr(tabdy), ptr(part)
                d.Mpad, d.nsplit, d.ntaps = mpad, nsplit, 4
                for i, (a, b) in enumerate(_TAPS2):
                    d.offx[i] = 0
                    d.offdy[i] = a * (dout.W + 2) + b
                _launch_wgrad(d, B * h * w, self.cin, self.cout, 4, ctx.code)
                ctx.wgrad_finish(part, sink.view(self.mod.weight), nsplit, 4, self.cout, self.cin, 1)

        if not WGRAD_LATE:
            weight_grad()
        if dx is not None:
            unit = consumer.u2 if (consumer is not None and consumer.pool_arg is None) else None
            slab = _igemm_bstat_slab(unit, consumer is not None and consumer.se is not None, B * h * w, self.cin, h * w, dx)
            _igemm(dout, dx, self.w.dgrad(), self.cin, h, w, 2, _TAPS2, 0,
                   stats=slab[0] if slab else None, bstat=slab[1] if slab else None)
            if slab:
                unit.bred_ready = True
        if WGRAD_LATE:
            weight_grad()


class OutConvPlan:
    """outc = Conv2d(64, num_classes, 1) (:125,162): NHWC activations -> NCHW fp32 logits."""

    def __init__(self, ctx: Ctx, mod: torch.nn.Conv2d, x: Act, name: str = "outc"):
        self.ctx, self.mod, self.x = ctx, mod, x
        self.K, self.cin = mod.out_channels, mod.in_channels
        self.nb = call("insar_conv1x1_out_bwd_blocks", x.B, x.H)
        self.cols = self.K * self.cin + self.K
        self.part = ctx.f32(self.nb, self.cols)
        self.folded = ctx.f32(self.cols)
        self.fused_src = None          # (unit, gate) when the last forward went through forward_fused
        self.part_red, self.reduce_rows = None, 0      # parameter-gradient partials written by the unit's reduce pass

    def params(self):
        return [self.mod.weight, self.mod.bias]

    def forward(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.fused_src = None
        x = self.x
        logits = out if out is not None else torch.empty((x.B, self.K, x.H, x.W), dtype=torch.float32, device=self.ctx.device)
        call("insar_conv1x1_out_fwd", x.ref, ptr(self.mod.weight),
             ptr(self.mod.bias) if self.mod.bias is not None else 0, ptr(logits), self.K, _lib.stream_ptr())
        return logits

    def forward_fused(self, unit: "ConvBN", gate: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """logits straight from the last unit's raw conv output (its BN/ReLU/gate pass and outc in one launch; the
        unit's output activation is never written)."""
        y = unit.y
        logits = out if out is not None else torch.empty((y.B, self.K, y.H, y.W), dtype=torch.float32, device=self.ctx.device)
        call("insar_bn_relu_apply_outc", y.ref, ptr(unit.scale), ptr(unit.shift), ptr(gate), ptr(self.mod.weight),
             ptr(self.mod.bias) if self.mod.bias is not None else 0, ptr(logits), self.K, 1, _lib.stream_ptr())
        self.fused_src = (unit, gate)
        return logits

    def backward(self, dlogits: torch.Tensor, sink: GradSink, dx: Optional[Act]) -> None:
        """dx = None: only outc's parameter gradients (side stream); the unit below recomputes its incoming gradient
        from dlogits (ConvBN.backward, outc_grad)."""
        def fold():
            self.ctx.colsum_into_grads(self.part, self.nb, self._grad_views(sink))

        if dx is None and self.reduce_rows:
            # the parameter-gradient partials come out of the unit's BatchNorm-backward reduce pass (same read of y): only
            # the fold is left, on the side stream, once that pass has been enqueued (UNetPlan.backward calls fold_fused)
            return
        if dx is None:
            if self.ctx.side is not None and not (PROFILER is not None and PROFILER.alone) and dlogits.is_cuda:
                if tape.REC is not None:
                    tape.REC.bad("record_stream on the caller's dlogits")      # this path is not replayable: stay eager
                if torch.cuda.is_current_stream_capturing():
                    self._keep = dlogits                  # a graph's private pool: keep the buffer alive instead
                else:
                    dlogits.record_stream(self.ctx.side)  # read on the side stream after the caller has dropped it
            with self.ctx.side_stream():
                if self.fused_src is not None:       # the input activation was never stored: recompute it from y
                    unit, gate = self.fused_src
                    call("insar_conv1x1_out_wgrad_y", unit.y.ref, ptr(unit.scale), ptr(unit.shift), ptr(gate),
                         ptr(self.mod.weight), ptr(dlogits), self.K, ptr(self.part), _lib.stream_ptr())
                else:
                    call("insar_conv1x1_out_wgrad", self.x.ref, ptr(self.mod.weight), ptr(dlogits), self.K,
                         ptr(self.part), _lib.stream_ptr())
                fold()
            return
        call("insar_conv1x1_out_bwd", self.x.ref, ptr(self.mod.weight), ptr(dlogits), self.K, dx.ref,
             ptr(self.part), _lib.stream_ptr())
        with self.ctx.side_stream():            # folds are off the critical path
            fold()

    def fused_grad(self, unit: "ConvBN"):
        """(gate, wpart) for the unit's reduce pass when outc's parameter gradients are to come out of it (OUTC_WGRAD_FUSE,
        the fused forward ran, K <= 2: the instantiation that has the registers), else None."""
        self.reduce_rows = 0
        if not (OUTC_WGRAD_FUSE and self.fused_src is not None and self.K <= 2 and self.fused_src[0] is unit):
            return None
        self.reduce_rows = unit.x.B * unit.red_rows
        if self.part_red is None or self.part_red.shape[0] != self.reduce_rows:
            self.part_red = self.ctx.f32(self.reduce_rows, self.cols)
        return self.fused_src[1], self.part_red

    def fold_fused(self, sink: GradSink) -> None:
        if not self.reduce_rows:
            return
        with self.ctx.side_stream():
            self.ctx.colsum_into_grads(self.part_red, self.reduce_rows, self._grad_views(sink))

    def _grad_views(self, sink: GradSink) -> List[torch.Tensor]:
        """outc's weight (and bias) gradient views; a bias-free outc still folds K trailing columns (into scratch)."""
        if self.mod.bias is not None:
            return [sink.view(self.mod.weight), sink.view(self.mod.bias)]
        return [sink.view(self.mod.weight), self.folded[self.K * self.cin:]]

    def virtual_grad_ok(self) -> bool:
        ch = 16 // self.ctx.esize
        return OUTC_FUSE and self.K <= 4 and self.cin % ch == 0 and 256 % (self.cin // ch) == 0


def pack_input(x: torch.Tensor, dst: Act) -> None:
    """NCHW tensor at the nn.Module boundary -> padded NHWC slice."""
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    call("insar_pack_nchw", ptr(x), dst.ref, _lib.stream_ptr())


def unpack_output(src: Act) -> torch.Tensor:
    out = torch.empty((src.B, src.c_len, src.H, src.W), dtype=torch.float32, device=src.buf.device)
    call("insar_unpack_nchw", src.ref, ptr(out), _lib.stream_ptr())
    return out


class UNetPlan(tape.PlanTape):
    """All buffers + the launch sequence of UNet.forward / backward for one input geometry."""
    per_stage_param_waits = True       # parallel.DataParallel(shard_optimizer=True): see _stage_gates

    def __init__(self, net, B: int, H: int, W: int, dtype: torch.dtype, device: torch.device):
        if H < 16 or W < 16:
            raise _lib.InsarError(f"H={H}, W={W}: four MaxPool2d(2) stages need at least 16 x 16 pixels")
        self.net, self.B, self.H, self.W = net, B, H, W
        ctx = self.ctx = Ctx(device, dtype)
        widths = [net.inc.double_conv[0].out_channels]
        downs = [net.down1, net.down2, net.down3, net.down4]
        for d in downs:
            widths.append(d[1].double_conv[0].out_channels)
        self.widths = widths
        cin = net.inc.double_conv[0].in_channels
        # nn.MaxPool2d(2) floors; when H or W is not a multiple of 16 some level is odd and the decoder's transposed conv
        # comes out one pixel short of its skip: the reference's bilinear-resize fallback (:138-139 ...), see UpPlan
        hs = [H >> l for l in range(5)]
        ws = [W >> l for l in range(5)]
        A = lambda l, c: Act.alloc(B, hs[l], ws[l], c, dtype, device)
        self.xin = A(0, cin)
        self.cat = [A(l, 2 * widths[l]) for l in range(4)]
        self.x5 = A(4, widths[4])
        self.pooled = [A(l + 1, widths[l]) for l in range(4)]
        self.dec = [A(l, widths[l]) for l in range(4)]
        # gradients
        self.dcat = [A(l, 2 * widths[l]) for l in range(4)]
        self.dx5 = A(4, widths[4])
        self.dpooled = [A(l + 1, widths[l]) for l in range(4)]
        self.ddec = [A(l, widths[l]) for l in range(4)]

        self.enc: List[DoubleConvPlan] = []
        enc_mods = [net.inc] + [d[1] for d in downs]
        enc_names = ["inc"] + [f"down{i}.1" for i in range(1, 5)]
        for l in range(5):
            xin = self.xin if l == 0 else self.pooled[l - 1]
            out = self.cat[l].slice(0, widths[l]) if l < 4 else self.x5
            self.enc.append(DoubleConvPlan(ctx, enc_mods[l], xin, out, enc_names[l]))
            if l < 4:
                self.enc[l].pool_out = self.pooled[l]
                ch = 16 // ctx.esize
                if POOL_FUSE and widths[l] % ch == 0 and 256 % (widths[l] // ch) == 0 and not ((hs[l] | ws[l]) & 1):
                    self.enc[l].pool_arg = torch.zeros((B, hs[l + 1], ws[l + 1], widths[l]), dtype=torch.uint8, device=device)
        ups = [net.up1, net.up2, net.up3, net.up4]
        convs = [net.conv1, net.conv2, net.conv3, net.conv4]
        self.up: List[UpPlan] = []
        self.dconv: List[DoubleConvPlan] = []
        for i in range(4):              # decoder stage i+1 works at level l = 3 - i
            l = 3 - i
            src = self.x5 if i == 0 else self.dec[l + 1]
            self.up.append(UpPlan(ctx, ups[i], src, self.cat[l].slice(widths[l], widths[l]), f"up{i + 1}"))
            self.dconv.append(DoubleConvPlan(ctx, convs[i], self.cat[l], self.dec[l], f"conv{i + 1}"))
        self.outc = OutConvPlan(ctx, net.outc, self.dec[0])
        # parameters in the order their gradients complete during backward, grouped by backward stage
        groups = grad_groups(net)
        self.grad_params = [p for g in groups for p in g]
        self.sink = GradSink(ctx, None, groups)
        # flat-buffer offsets at which each backward stage's gradients are complete (for DP buckets)
        self.stage_sizes = self.sink.group_sizes
        self.stage_ends = [sum(self.stage_sizes[:i + 1]) for i in range(len(self.stage_sizes))]
        self._closes = {}
        self._gate_events = None
        self.busy = False
        self.training = True
        gws = [b.u1.w for b in self.enc + self.dconv if b.u1.w is not None] + [b.u2.w for b in self.enc + self.dconv]
        gws += [u.w for u in self.up]
        self.weightset = WeightSet(ctx, gws)
        self.bn_modules = [u.bn for b in self.enc + self.dconv for u in (b.u1, b.u2)]
        self._tape_setup()

    def bucket_closes(self, min_elems: int):
        if min_elems not in self._closes:
            from .parallel import plan_buckets
            self._closes[min_elems] = set(plan_buckets(self.stage_sizes, min_elems))
        return self._closes[min_elems]

    # ---- forward ----------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        """The ordinary launch sequence (_forward_eager) or, in the steady state of a training loop, its launch tape (tape.py)."""
        if not (self._tape_allowed(training, not self.net._hooks.get("param_waits")) and self.outc.virtual_grad_ok()):
            return self._forward_eager(x, training)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        logits = torch.empty((self.B, self.outc.K, self.H, self.W), dtype=torch.float32, device=self.ctx.device)
        slots = {"x": x.data_ptr(), "logits": logits.data_ptr()}
        box = {}

        def eager():
            box["out"] = self._forward_eager(x, training)
            return box["out"]

        # a recording's own logits tensor (allocated inside the eager code) becomes the "logits" slot of the tape
        out, replayed = self._run(self._tape_key("f"), eager, slots, {x.data_ptr(): "x"},
                                  dyn_after=lambda o: {o.data_ptr(): "logits"})
        if replayed:
            self.training = training
            return logits
        return out

    def _stage_gates(self, waits):
        """Sharded data parallelism (parallel.DataParallel(shard_optimizer=True), ShardedAdam(defer_gather=True)): the
        parameter all-gathers of the last optimizer step may still be in flight, issued bucket by bucket in the order this
        forward pass needs them. On the side stream, in forward order of the backward stages: wait for the stage's bucket
        (`waits(stage)`), re-lay THAT stage's GEMM weight copies, record an event. Returns gate(stage): the main stream
        waits for the stage's event right before the first launch that reads the stage's parameters — the first encoder
        levels run while the decoder's 58 MB are still arriving. (The reference has no distributed code)"""
        sets = self.weightset.stage_sets(self.sink.groups)
        n = len(self.sink.groups)
        if self._gate_events is None:
            self._gate_events = [torch.cuda.Event() for _ in range(n)]
        with self.ctx.side_stream():
            for st in reversed(range(n)):                # forward order: enc0 = last backward stage ... decoder stage 3 + outc = stage 0
                waits(st)
                if sets[st] is not None:
                    sets[st].refresh()
                self._gate_events[st].record()
        self.ctx._side_busy = False          # the per-stage events order the main stream; a join would wait for EVERY bucket
        evs = self._gate_events
        return lambda st: torch.cuda.current_stream().wait_event(evs[st])

    def _forward_eager(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        s = _lib.stream_ptr
        self.training = training
        waits = self.net._hooks.get("param_waits")
        gate = None
        # all GEMM-layout weight copies in one launch when the masters moved. The first layer works on the fp32
        # masters, so the re-layout runs on the side stream beside input packing and the first conv / BN / ReLU
        # passes; the first GEMM conv (ConvBN.forward_conv) joins it.
        if waits is not None and self.ctx.side is not None:
            gate = self._stage_gates(waits)
        elif PREP_SIDE:
            if waits is not None:
                for st in range(len(self.sink.groups)):
                    waits(st)
            with self.ctx.side_stream():
                self.weightset.refresh()
        else:
            self.weightset.refresh()
        pack_input(x, self.xin)
        sync = self.net._hooks.get("sync_bn") if training else None       # (process group, world) under DataParallel(sync_bn=True)
        for l in range(5):
            if gate:
                gate(8 - l)
            self.enc[l].forward(training, sync=sync)          # levels 0-3 write their max-pool too (pool_out)
        for i in range(3):
            if gate:
                gate(3 - i)
            self.up[i].forward()
            self.dconv[i].forward(training, sync=sync)
        if gate:
            gate(0)
        self.up[3].forward()
        if self.outc.virtual_grad_ok():      # last block: BN/ReLU/gate + outc in one pass, no 64-channel output tensor
            return self.dconv[3].forward(training, self.outc, sync=sync)
        self.dconv[3].forward(training, sync=sync)
        return self.outc.forward()

    # ---- backward ---------------------------------------------------------------------------------
    def backward(self, dlogits: torch.Tensor, on_bucket=None) -> List[torch.Tensor]:
        """The ordinary launch sequence (_backward_eager) or its launch tape; data-parallel hooks keep the ordinary code."""
        if dlogits.dtype != torch.float32 or not dlogits.is_contiguous():
            dlogits = dlogits.float().contiguous()
        self.sink.select()
        # (a master that moved between forward and backward — an optimizer step in between — is re-laid per weight by the
        # ordinary code; a tape would keep the copies the forward used)
        if not (self._tape_allowed(self.training, on_bucket is None and not self.weightset.stale()) and self.outc.virtual_grad_ok()):
            return self._backward_eager(dlogits, on_bucket)
        out, replayed = self._run(self._tape_key("b"), lambda: self._backward_eager(dlogits, None),
                                  {"dlogits": dlogits.data_ptr()}, {dlogits.data_ptr(): "dlogits"})
        if replayed:
            return [self.sink.view(p) for p in self.grad_params]
        return out

    def _backward_eager(self, dlogits: torch.Tensor, on_bucket=None) -> List[torch.Tensor]:
        s = _lib.stream_ptr
        sink, training, w = self.sink, self.training, self.widths
        fuse = self.outc.virtual_grad_ok()
        wg = self.outc.fused_grad(self.dconv[3].u2) if fuse else None
        self.outc.backward(dlogits, sink, None if fuse else self.ddec[0])
        for i in (3, 2, 1, 0):
            l = 3 - i
            og = (dlogits, self.net.outc.weight.detach(), self.outc.K) if (fuse and i == 3) else None
            if og and wg:
                og = og + wg
            self.dconv[i].backward(None if og else self.ddec[l], sink, training, self.dcat[l], og)
            if og and wg:
                self.outc.fold_fused(sink)
            dsrc = self.dx5 if i == 0 else self.ddec[l + 1]
            self.up[i].backward(self.dcat[l].slice(w[l], w[l]), sink, dsrc, self.enc[4] if i == 0 else self.dconv[i - 1])
            if on_bucket is not None:
                on_bucket(self, ("dec", i))
        for l in (4, 3, 2, 1, 0):
            dout = self.dx5 if l == 4 else self.dcat[l].slice(0, w[l])
            # levels 0-3: the gradient through MaxPool2d(2) (dpooled[l], written by the level below) joins the skip
            # gradient inside this block's BatchNorm-backward passes when the forward pass kept the arg-max map
            pg = (self.dpooled[l], self.enc[l].pool_arg) if (l < 4 and self.enc[l].pool_arg is not None) else None
            self.enc[l].backward(dout, sink, training, self.dpooled[l - 1] if l > 0 else None, None, pg)
            if l > 0 and self.enc[l - 1].pool_arg is None:
                call("insar_maxpool2_bwd", self.enc[l - 1].out.ref, self.dpooled[l - 1].ref,
                     self.dcat[l - 1].slice(0, w[l - 1]).ref, 1, s())
            if on_bucket is not None:
                on_bucket(self, ("enc", l))
        self.ctx.join_side()
        return [sink.view(p) for p in self.grad_params]
