"""Loss entry points of the training loop, on the HIP path.

`CrossEntropyLoss(ignore_index=255)` is the drop-in for the reference's criterion
(Unet-ChannalAttention.py:465, used at :344): mean over non-ignored pixels of
-log softmax(logits)[target]; forward and the gradient are produced by one fused pass.
`DiceLoss` / `DiceCELoss` are build-side additions (the reference has no Dice loss,
SURVEY §0): standard soft-Dice on softmax probabilities.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr


def _prep(logits: torch.Tensor, target: torch.Tensor, who: str):
    if not logits.is_cuda:
        raise _lib.InsarError(f"{who}: logits are on {logits.device}; the HIP path needs a ROCm tensor (no CPU fallback)")
    if logits.dim() < 2:
        raise _lib.InsarError(f"{who}: logits must be [B, K, ...]")
    B, K = logits.shape[0], logits.shape[1]
    HW = 1
    for d in logits.shape[2:]:
        HW *= d
    if target.shape != (B,) + tuple(logits.shape[2:]):
        raise _lib.InsarError(f"{who}: target shape {tuple(target.shape)} does not match logits {tuple(logits.shape)}")
    lg = logits.detach()
    if lg.dtype != torch.float32 or not lg.is_contiguous():
        lg = lg.float().contiguous()
    tg = target
    if tg.dtype != torch.int64 or not tg.is_contiguous():
        tg = tg.long().contiguous()
    return lg, tg, B, K, HW


def _scale_by(dl: torch.Tensor, g: torch.Tensor, in_dtype) -> torch.Tensor:
    """d loss / d logits times the incoming gradient of the scalar loss, on the HIP path (insar_mul_dev_f32: the factor is
    read from device memory; dl itself is kept for a second backward)."""
    if g.numel() != 1 or not g.is_cuda:
        raise _lib.InsarError("loss backward: the incoming gradient must be a device scalar")
    gs = g.detach()
    if gs.dtype != torch.float32:
        gs = gs.float()
    out = torch.empty_like(dl)
    call("insar_mul_dev_f32", ptr(out), ptr(dl), dl.numel(), ptr(gs), _lib.stream_ptr())
    return out if in_dtype == torch.float32 else out.to(in_dtype)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        lg, tg, B, K, HW = _prep(logits, target, "CrossEntropyLoss")
        dl = torch.empty_like(lg)
        nb = call("insar_ce_blocks", B * HW)
        ws = torch.empty(2 + 2 * nb, dtype=torch.float32, device=lg.device)
        out = torch.empty(1, dtype=torch.float32, device=lg.device)
        call("insar_cross_entropy", ptr(lg), ptr(tg), B, K, HW, ignore_index, ptr(dl), ptr(out), ptr(ws), _lib.stream_ptr())
        ctx.dl = dl
        ctx.in_dtype = logits.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return _scale_by(ctx.dl, g, ctx.in_dtype), None, None


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, smooth):
        lg, tg, B, K, HW = _prep(logits, target, "DiceLoss")
        dl = torch.empty_like(lg)
        nb = call("insar_ce_blocks", B * HW)
        ws = torch.empty(3 * K * (nb + 1), dtype=torch.float32, device=lg.device)
        out = torch.empty(1, dtype=torch.float32, device=lg.device)
        call("insar_dice", ptr(lg), ptr(tg), B, K, HW, ignore_index, float(smooth), ptr(dl), ptr(out), ptr(ws),
             _lib.stream_ptr())
        ctx.dl = dl
        ctx.in_dtype = logits.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return _scale_by(ctx.dl, g, ctx.in_dtype), None, None, None


class _DiceCEFn(torch.autograd.Function):
    """ce_weight*CE + dice_weight*Dice: one statistics pass and one gradient pass over the logits."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index, smooth, ce_weight, dice_weight):
        lg, tg, B, K, HW = _prep(logits, target, "DiceCELoss")
        dl = torch.empty_like(lg)
        nb = call("insar_ce_blocks", B * HW)
        ws = torch.empty(3 + 3 * K + nb * (2 + 3 * K), dtype=torch.float32, device=lg.device)
        out = torch.empty(3, dtype=torch.float32, device=lg.device)
        call("insar_dice_ce", ptr(lg), ptr(tg), B, K, HW, ignore_index, float(smooth), float(ce_weight), float(dice_weight),
             ptr(dl), ptr(out), ptr(ws), _lib.stream_ptr())
        ctx.dl = dl
        ctx.in_dtype = logits.dtype
        ctx.parts = out           # [combined, ce, dice] for logging
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return _scale_by(ctx.dl, g, ctx.in_dtype), None, None, None, None, None


class CrossEntropyLoss(nn.Module):
    """Drop-in for nn.CrossEntropyLoss(ignore_index=...) with mean reduction."""

    def __init__(self, weight=None, ignore_index: int = -100, reduction: str = "mean", label_smoothing: float = 0.0):
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0.0:
            raise _lib.InsarError("CrossEntropyLoss HIP path: only weight=None, reduction='mean', label_smoothing=0")
        self.ignore_index = ignore_index

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _CEFn.apply(logits, target, self.ignore_index)


class DiceLoss(nn.Module):
    """1 - mean_c (2*I_c + smooth) / (P_c + T_c + smooth) on softmax probabilities."""

    def __init__(self, ignore_index: int = 255, smooth: float = 1.0):
        super().__init__()
        self.ignore_index, self.smooth = ignore_index, smooth

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _DiceFn.apply(logits, target, self.ignore_index, self.smooth)


class DiceCELoss(nn.Module):
    """ce_weight * CE + dice_weight * Dice (the 'Dice+CE' training objective of config 2)."""

    def __init__(self, ignore_index: int = 255, smooth: float = 1.0, ce_weight: float = 1.0, dice_weight: float = 1.0):
        super().__init__()
        self.ce = CrossEntropyLoss(ignore_index=ignore_index)
        self.dice = DiceLoss(ignore_index=ignore_index, smooth=smooth)
        self.ce_weight, self.dice_weight = ce_weight, dice_weight

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _DiceCEFn.apply(logits, target, self.ce.ignore_index, self.dice.smooth, self.ce_weight, self.dice_weight)
