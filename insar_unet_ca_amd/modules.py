"""Drop-in nn.Module surface of the reference's U-Net-CA (Unet-ChannalAttention.py:45-163).

Same class names, constructor signatures, attribute tree and state_dict as the
reference's `SELayer`, `DoubleConv` and `UNet`; parameters live in ordinary
nn.Conv2d / nn.BatchNorm2d / nn.Linear / nn.ConvTranspose2d containers (so `.pth` files
interchange and torch's default initialisation is consumed in the reference's order), but
`forward` runs the hand-written HIP kernels through the C ABI. There is no CPU or eager
fallback: calling `forward` with a non-ROCm tensor raises.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import InsarBnSeBwd, InsarCam, InsarSeFwd, call, ptr
from . import engine
from .engine import Act, Ctx, DoubleConvPlan, GradSink, UNetPlan, pack_input, unpack_output


def _require_device(x: torch.Tensor, who: str) -> None:
    if not x.is_cuda:
        raise _lib.InsarError(
            f"{who}: input is on {x.device}; the HIP path needs a ROCm device tensor (no CPU fallback). "
            "Move the model and data to 'cuda'.")


def _resolve_dtype(module) -> torch.dtype:
    dt = getattr(module, "compute_dtype", None)
    if dt is None:
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
        else:
            dt = torch.float32
    if dt not in (torch.float32, torch.bfloat16):
        raise _lib.InsarError(f"compute dtype {dt} not supported (float32 | bfloat16)")
    return dt


class _Lease:
    """Marks a plan busy between a grad-enabled forward and its backward (released on GC too)."""

    def __init__(self, plan):
        self.plan = plan
        plan.busy = True

    def release(self):
        if self.plan is not None:
            self.plan.busy = False
            self.plan = None

    def __del__(self):
        self.release()


class _PlanCache:
    """Plans (preallocated buffers + launch sequence) per input geometry. A plan for config 2 holds ~6 GB, so
    the cache keeps at most `MAX_KEYS` geometries and drops the least recently used idle ones beyond that
    (a loader with a ragged last batch needs two; variable tile sizes must not grow memory without bound)."""

    MAX_KEYS = 4

    def __init__(self):
        self.plans: "collections.OrderedDict[tuple, list]" = collections.OrderedDict()

    def get(self, key, factory):
        lst = self.plans.get(key)
        if lst is None:
            lst = self.plans[key] = []
            for old in list(self.plans.keys()):
                if len(self.plans) <= self.MAX_KEYS:
                    break
                if old != key and not any(p.busy for p in self.plans[old]):
                    del self.plans[old]
        self.plans.move_to_end(key)
        for p in lst:
            if not p.busy:
                return p
        p = factory()
        lst.append(p)
        return p

    def clear(self):
        self.plans.clear()


# ------------------------------------------------------------------------------------------------
# SELayer
# ------------------------------------------------------------------------------------------------
class _SEPlan:
    """Stand-alone SELayer on a raw tensor (no BN/ReLU in front): out = x * gate(mean_hw x)."""

    def __init__(self, mod, B, Cn, H, W, dtype, device):
        self.mod = mod
        ctx = self.ctx = Ctx(device, dtype)
        if Cn % 64:
            raise _lib.InsarError(f"SELayer: channel={Cn} must be a multiple of 64 on the HIP path")
        self.B, self.C, self.H, self.W = B, Cn, H, W
        self.x = Act.alloc(B, H, W, Cn, dtype, device)
        self.out = Act.alloc(B, H, W, Cn, dtype, device)
        self.dout = Act.alloc(B, H, W, Cn, dtype, device)
        self.dx = Act.alloc(B, H, W, Cn, dtype, device)
        self.cr = mod.fc[0].out_features
        self.rpp = engine._rows_per_part(B, H)
        self.rows = -(-H // self.rpp)
        self.part = ctx.f32(B * self.rows, 2, Cn)
        self.pooled, self.red_part = ctx.f32(B, 2, Cn), ctx.f32(B * self.rows, 2, Cn)
        self.sq, self.gate, self.coefB = ctx.f32(B, Cn), ctx.f32(B, Cn), ctx.f32(B, Cn)
        self.hid = ctx.f32(B, self.cr)
        self.ones, self.zeros = ctx.const(1.0, Cn), ctx.const(0.0, Cn)
        self.k1, self.k2, self.scratch = ctx.f32(Cn), ctx.f32(Cn), ctx.f32(2, Cn)
        self.ws = ctx.f32(B * (3 * Cn + self.cr))
        self.sink = GradSink(ctx, [mod.fc[0].weight, mod.fc[2].weight])
        self.busy = False

    def forward(self, x):
        s = _lib.stream_ptr()
        pack_input(x, self.x)
        call("insar_se_squeeze", self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.part), 0, self.rpp, s)
        d = InsarSeFwd()
        d.part, d.rows, d.pooled = ptr(self.part), self.rows, ptr(self.pooled)
        d.B, d.H, d.W, d.C, d.Cr = self.B, self.H, self.W, self.C, self.cr
        d.scale, d.shift = ptr(self.ones), ptr(self.zeros)
        d.w1, d.w2 = ptr(self.mod.fc[0].weight), ptr(self.mod.fc[2].weight)
        d.sq, d.hid, d.gate = ptr(self.sq), ptr(self.hid), ptr(self.gate)
        call("insar_se_excite", C.byref(d), s)
        call("insar_bn_relu_apply", self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.gate), self.out.ref, 0, s)
        return unpack_output(self.out)

    def backward(self, g):
        s = _lib.stream_ptr()
        self.sink.select()
        pack_input(g, self.dout)
        call("insar_bnrelu_bwd_reduce", self.dout.ref, self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.red_part), 0, self.rpp, s)
        d = InsarBnSeBwd()
        d.B, d.H, d.W, d.C, d.Cr, d.use_se = self.B, self.H, self.W, self.C, self.cr, 1
        d.mean, d.invstd = ptr(self.zeros), ptr(self.ones)
        d.pooled, d.sq, d.hid, d.gate = ptr(self.pooled), ptr(self.sq), ptr(self.hid), ptr(self.gate)
        w1, w2 = self.mod.fc[0].weight, self.mod.fc[2].weight
        d.w1, d.w2 = ptr(w1), ptr(w2)
        d.dw1, d.dw2 = ptr(self.sink.view(w1)), ptr(self.sink.view(w2))
        d.dgamma, d.dbeta = ptr(self.scratch[0]), ptr(self.scratch[1])
        d.coefB, d.k1, d.k2 = ptr(self.coefB), ptr(self.k1), ptr(self.k2)
        call("insar_bnse_bwd_coef", C.byref(d), ptr(self.red_part), self.rows, ptr(self.ones), ptr(self.zeros), ptr(self.ws), 0, 0, s)
        call("insar_bnrelu_bwd_apply", self.dout.ref, self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.zeros),
             ptr(self.ones), ptr(self.gate), ptr(self.coefB), ptr(self.k1), ptr(self.k2), self.dx.ref, 0, s)
        return unpack_output(self.dx), self.sink.view(w1), self.sink.view(w2)


class _SEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, track, x, w1, w2):
        out = plan.forward(x)
        ctx.plan = plan
        ctx.lease = _Lease(plan) if track else None
        return out

    @staticmethod
    def backward(ctx, g):
        dx, dw1, dw2 = ctx.plan.backward(g)
        if ctx.lease:
            ctx.lease.release()
        return None, None, dx, dw1, dw2


class SELayer(nn.Module):
    """Squeeze-and-Excitation channel attention (Unet-ChannalAttention.py:45-72)."""

    def __init__(self, channel: int, reduction: int = 16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(
            nn.Linear(channel, channel // reduction, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(channel // reduction, channel, bias=False),
            nn.Sigmoid(),
        )
        self.compute_dtype: Optional[torch.dtype] = None
        self._plans = _PlanCache()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_device(x, "SELayer")
        b, c, h, w = x.shape
        dt = _resolve_dtype(self)
        plan = self._plans.get((b, c, h, w, dt, x.device), lambda: _SEPlan(self, b, c, h, w, dt, x.device))
        return _SEFn.apply(plan, torch.is_grad_enabled(), x, self.fc[0].weight, self.fc[2].weight)


# ------------------------------------------------------------------------------------------------
# ChannelAttentionModule (config 5)
# ------------------------------------------------------------------------------------------------
class _CamPlan:
    """out = x * sigmoid(MLP(avg_hw x) + MLP(max_hw x)) on a raw tensor (csrc/cam.hip)."""

    def __init__(self, mod, B, Cn, H, W, dtype, device):
        self.mod = mod
        ctx = self.ctx = Ctx(device, dtype)
        if Cn % 64:
            raise _lib.InsarError(f"ChannelAttentionModule: in_channels={Cn} must be a multiple of 64 on the HIP path")
        self.B, self.C, self.H, self.W = B, Cn, H, W
        self.cr = mod.mlp[0].out_channels
        self.x = Act.alloc(B, H, W, Cn, dtype, device)
        self.out = Act.alloc(B, H, W, Cn, dtype, device)
        self.dout = Act.alloc(B, H, W, Cn, dtype, device)
        self.dx = Act.alloc(B, H, W, Cn, dtype, device)
        self.rpp = engine._rows_per_part(B, H)
        self.rows = -(-H // self.rpp)
        self.psum, self.pmax = ctx.f32(B * self.rows, Cn), ctx.f32(B * self.rows, Cn)
        self.parg = torch.zeros(B * self.rows, Cn, dtype=torch.int32, device=device)
        self.avg, self.mx, self.gate = ctx.f32(B, Cn), ctx.f32(B, Cn), ctx.f32(B, Cn)
        self.arg = torch.zeros(B, Cn, dtype=torch.int32, device=device)
        self.ha, self.hm = ctx.f32(B, self.cr), ctx.f32(B, self.cr)
        self.coefB, self.dmax = ctx.f32(B, Cn), ctx.f32(B, Cn)
        self.red_part = ctx.f32(B * self.rows, 2, Cn)
        self.ws = ctx.f32(B * (Cn + 2 * self.cr))
        self.ones, self.zeros = ctx.const(1.0, Cn), ctx.const(0.0, Cn)
        self.sink = GradSink(ctx, [mod.mlp[0].weight, mod.mlp[2].weight])
        self.busy = False

    def _desc(self):
        d = InsarCam()
        d.B, d.H, d.W, d.C, d.Cr, d.rows = self.B, self.H, self.W, self.C, self.cr, self.rows
        d.psum, d.pmax, d.parg = ptr(self.psum), ptr(self.pmax), ptr(self.parg)
        d.w1, d.w2 = ptr(self.mod.mlp[0].weight), ptr(self.mod.mlp[2].weight)   # (Cr,C,1,1), (C,Cr,1,1)
        d.avg, d.mx, d.arg = ptr(self.avg), ptr(self.mx), ptr(self.arg)
        d.ha, d.hm, d.gate = ptr(self.ha), ptr(self.hm), ptr(self.gate)
        d.coefB, d.dmax, d.ws = ptr(self.coefB), ptr(self.dmax), ptr(self.ws)
        d.accumulate = 0
        return d

    def forward(self, x):
        s = _lib.stream_ptr()
        pack_input(x, self.x)
        call("insar_cam_pool", self.x.ref, ptr(self.psum), ptr(self.pmax), ptr(self.parg), self.rpp, s)
        d = self._desc()
        call("insar_cam_excite", C.byref(d), s)
        call("insar_bn_relu_apply", self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.gate), self.out.ref, 0, s)
        return unpack_output(self.out)

    def backward(self, g):
        s = _lib.stream_ptr()
        self.sink.select()
        pack_input(g, self.dout)
        call("insar_bnrelu_bwd_reduce", self.dout.ref, self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.red_part), 0, self.rpp, s)
        w1, w2 = self.mod.mlp[0].weight, self.mod.mlp[2].weight
        d = self._desc()
        d.dw1, d.dw2 = ptr(self.sink.view(w1)), ptr(self.sink.view(w2))
        call("insar_cam_bwd_coef", C.byref(d), ptr(self.red_part), self.rows, s)
        call("insar_bnrelu_bwd_apply", self.dout.ref, self.x.ref, ptr(self.ones), ptr(self.zeros), ptr(self.zeros),
             ptr(self.ones), ptr(self.gate), ptr(self.coefB), ptr(self.zeros), ptr(self.zeros), self.dx.ref, 0, s)
        call("insar_cam_scatter_max", self.dx.ref, ptr(self.dmax), ptr(self.arg), s)
        return unpack_output(self.dx), self.sink.view(w1), self.sink.view(w2)


class ChannelAttentionModule(nn.Module):
    """Channel attention of the DeepLabV3 variant (DeepLabV3-ChannelAttention.py:49-79): same constructor,
    attribute names (`avg_pool`, `max_pool`, `mlp`, `sigmoid`) and state_dict (`mlp.0.weight`, `mlp.2.weight`)."""

    def __init__(self, in_channels: int, reduction_ratio: int = 16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.mlp = nn.Sequential(
            nn.Conv2d(in_channels, in_channels // reduction_ratio, 1, bias=False),
            nn.ReLU(),
            nn.Conv2d(in_channels // reduction_ratio, in_channels, 1, bias=False),
        )
        self.sigmoid = nn.Sigmoid()
        self.compute_dtype: Optional[torch.dtype] = None
        self._plans = _PlanCache()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_device(x, "ChannelAttentionModule")
        b, c, h, w = x.shape
        dt = _resolve_dtype(self)
        plan = self._plans.get((b, c, h, w, dt, x.device), lambda: _CamPlan(self, b, c, h, w, dt, x.device))
        return _SEFn.apply(plan, torch.is_grad_enabled(), x, self.mlp[0].weight, self.mlp[2].weight)


# ------------------------------------------------------------------------------------------------
# DoubleConv
# ------------------------------------------------------------------------------------------------
class _DoubleConvRunner:
    def __init__(self, mod, B, H, W, dtype, device):
        ctx = self.ctx = Ctx(device, dtype)
        cin = mod.double_conv[0].in_channels
        cout = mod.double_conv[3].out_channels
        self.x = Act.alloc(B, H, W, cin, dtype, device)
        self.out = Act.alloc(B, H, W, cout, dtype, device)
        self.dout = Act.alloc(B, H, W, cout, dtype, device)
        self.dx = Act.alloc(B, H, W, cin, dtype, device) if cin > 4 else None
        self.plan = DoubleConvPlan(ctx, mod, self.x, self.out, "double_conv")
        self.params = self.plan.params()
        self.sink = GradSink(ctx, self.params)
        self.busy = False
        self.training = True

    def forward(self, x, training):
        self.training = training
        pack_input(x, self.x)
        self.plan.forward(training)
        return unpack_output(self.out)

    def backward(self, g, need_dx):
        self.sink.select()
        pack_input(g, self.dout)
        if need_dx and self.dx is None:
            raise _lib.InsarError("DoubleConv: input gradient is not provided for in_channels <= 4 (first layer)")
        self.plan.backward(self.dout, self.sink, self.training, self.dx if need_dx else None)
        self.ctx.join_side()
        dx = unpack_output(self.dx) if need_dx else None
        return dx, [self.sink.view(p) for p in self.params]


class _DoubleConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, training, track, x, *params):
        out = runner.forward(x, training)
        ctx.runner = runner
        ctx.need_dx = x.requires_grad
        ctx.lease = _Lease(runner) if track else None
        return out

    @staticmethod
    def backward(ctx, g):
        dx, grads = ctx.runner.backward(g, ctx.need_dx)
        if ctx.lease:
            ctx.lease.release()
        return (None, None, None, dx) + tuple(grads)


class DoubleConv(nn.Module):
    """(Conv3x3 -> BN -> ReLU) x 2 with optional SE (Unet-ChannalAttention.py:75-97)."""

    def __init__(self, in_channels: int, out_channels: int, use_se: bool = False):
        super().__init__()
        layers = [
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        ]
        if use_se:
            layers.append(SELayer(out_channels))
        self.double_conv = nn.Sequential(*layers)
        self.compute_dtype: Optional[torch.dtype] = None
        self._plans = _PlanCache()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_device(x, "DoubleConv")
        b, _, h, w = x.shape
        dt = _resolve_dtype(self)
        runner = self._plans.get((b, h, w, dt, x.device), lambda: _DoubleConvRunner(self, b, h, w, dt, x.device))
        return _DoubleConvFn.apply(runner, self.training, torch.is_grad_enabled(), x, *runner.params)


# ------------------------------------------------------------------------------------------------
# MaxPool2d(2) (stand-alone use of `down_i[0]`; inside UNet.forward the pool runs in the plan)
# ------------------------------------------------------------------------------------------------
class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        b, c, h, w = x.shape
        if h % 2 or w % 2 or c % 8:
            raise _lib.InsarError("MaxPool2d(2) HIP path: H, W even and C a multiple of 8 required")
        xa = Act.alloc(b, h, w, c, dtype, x.device)
        ya = Act.alloc(b, h // 2, w // 2, c, dtype, x.device)
        pack_input(x, xa)
        call("insar_maxpool2_fwd", xa.ref, ya.ref, _lib.stream_ptr())
        ctx.xa, ctx.dtype = xa, dtype
        return unpack_output(ya)

    @staticmethod
    def backward(ctx, g):
        xa = ctx.xa
        ga = Act.alloc(xa.B, xa.H // 2, xa.W // 2, xa.C, ctx.dtype, g.device)
        dxa = Act.alloc(xa.B, xa.H, xa.W, xa.C, ctx.dtype, g.device)
        pack_input(g, ga)
        call("insar_maxpool2_bwd", xa.ref, ga.ref, dxa.ref, 0, _lib.stream_ptr())
        return unpack_output(dxa), None


class MaxPool2d(nn.MaxPool2d):
    """nn.MaxPool2d(2) whose forward/backward run the HIP kernels (:106-109)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_device(x, "MaxPool2d")
        ks = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        if ks != 2 or self.padding not in (0, (0, 0)) or self.dilation not in (1, (1, 1)):
            raise _lib.InsarError("only MaxPool2d(2) is part of the HIP path")
        return _PoolFn.apply(x, torch.float32)


# ------------------------------------------------------------------------------------------------
# UNet
# ------------------------------------------------------------------------------------------------
class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, training, track, hooks, x, *params):
        if x.requires_grad:
            # nn.Module semantics would return d(loss)/d(input); the first layer's input gradient is not part of the
            # hot path (the reference never asks for it: images come from the loader). Say so instead of returning None.
            raise _lib.InsarError("UNet: the gradient with respect to the input tensor is not provided by the HIP path; "
                                  "pass the images with requires_grad=False")
        if hooks and hooks.get("before_forward"):
            hooks["before_forward"](plan)          # DataParallel: parameter all-gathers still in flight
        logits = plan.forward(x, training)
        ctx.plan, ctx.hooks = plan, hooks
        ctx.lease = _Lease(plan) if track else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        hooks = ctx.hooks or {}
        if hooks.get("on_begin"):
            hooks["on_begin"](plan)
        on_bucket = hooks.get("on_bucket")
        early = hooks.get("on_stage_optim")
        if early is not None and on_bucket is None:
            # optimizer fused into backward (optim.Adam.fuse_into_backward): after every backward stage its parameters are
            # updated and re-laid out on the side stream. Not under data parallelism (the gradients are not final yet there).
            stage = [0]
            early(plan, -1)

            def on_bucket(pl, tag):
                early(pl, stage[0])
                stage[0] += 1
        grads = plan.backward(dlogits, on_bucket=on_bucket)
        if hooks.get("on_done"):
            hooks["on_done"](plan)
        if ctx.lease:
            ctx.lease.release()
        needs = ctx.needs_input_grad[5:]
        mode = hooks.get("grad_mode", "direct" if hooks.get("direct_grad", True) else "autograd") if ctx.hooks is not None else "autograd"
        if mode == "none":          # sharded optimizer: the reduced gradients live in its shards, p.grad stays None
            return (None,) * (5 + len(grads))
        if mode == "direct":
            # Hand the flat-buffer views to the parameters ourselves (what AccumulateGrad would do,
            # minus one 125 MB clone per step): first gradient -> alias the view, otherwise add.
            # Frozen parameters (requires_grad=False) get no .grad, as with AccumulateGrad.
            for p, g, need in zip(plan.grad_params, grads, needs):
                if not need:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
            return (None,) * (5 + len(grads))
        return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(grads, needs))


class UNet(nn.Module):
    """U-Net with optional SE channel attention (Unet-ChannalAttention.py:100-163).

    forward(x: [B, in_channels, H, W]) -> logits [B, num_classes, H, W] (float32, NCHW).
    `compute_dtype` (None -> float32, or torch.bfloat16; also follows torch.autocast) selects the
    arithmetic of the HIP kernels; parameters stay float32 masters either way.
    """

    def __init__(self, in_channels: int = 1, num_classes: int = 2, use_se: bool = False,
                 compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.inc = DoubleConv(in_channels, 64, use_se=use_se)
        self.down1 = nn.Sequential(MaxPool2d(2), DoubleConv(64, 128, use_se=use_se))
        self.down2 = nn.Sequential(MaxPool2d(2), DoubleConv(128, 256, use_se=use_se))
        self.down3 = nn.Sequential(MaxPool2d(2), DoubleConv(256, 512, use_se=use_se))
        self.down4 = nn.Sequential(MaxPool2d(2), DoubleConv(512, 1024, use_se=use_se))
        self.up1 = nn.ConvTranspose2d(1024, 512, kernel_size=2, stride=2)
        self.conv1 = DoubleConv(1024, 512, use_se=use_se)
        self.up2 = nn.ConvTranspose2d(512, 256, kernel_size=2, stride=2)
        self.conv2 = DoubleConv(512, 256, use_se=use_se)
        self.up3 = nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2)
        self.conv3 = DoubleConv(256, 128, use_se=use_se)
        self.up4 = nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2)
        self.conv4 = DoubleConv(128, 64, use_se=use_se)
        self.outc = nn.Conv2d(64, num_classes, kernel_size=1)
        self.compute_dtype = compute_dtype
        self._plans = _PlanCache()
        self._hooks: dict = {}
        self.per_stage_param_waits = True      # its plans wait for sharded-DP parameter buckets stage by stage (engine.UNetPlan)

    def set_compute_dtype(self, dtype: Optional[torch.dtype]) -> "UNet":
        self.compute_dtype = dtype
        return self

    def _plan(self, x: torch.Tensor) -> UNetPlan:
        b, c, h, w = x.shape
        if c != self.inc.double_conv[0].in_channels:
            raise _lib.InsarError(f"UNet: expected {self.inc.double_conv[0].in_channels} input channels, got {c}")
        dt = _resolve_dtype(self)
        return self._plans.get((b, h, w, dt, x.device), lambda: UNetPlan(self, b, h, w, dt, x.device))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_device(x, "UNet")
        plan = self._plan(x)
        # BatchNorm mode is a per-layer flag in nn.Module; the plan runs the whole net in one mode. A frozen encoder
        # (`net.down1.eval()` under `net.train()`) would silently be ignored: refuse it.
        for bn in plan.bn_modules:
            if bn.training != self.training:
                raise _lib.InsarError("UNet: mixed BatchNorm modes (a sub-module's .training differs from the net's) "
                                      "are not supported by the HIP path; call net.train() / net.eval() on the whole net")
        return _UNetFn.apply(plan, self.training, torch.is_grad_enabled(), self._hooks, x, *plan.grad_params)

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda()/.float() may move parameters: cached plans hold raw pointers -> drop them
        out = super()._apply(fn, *args, **kwargs)
        for m in self.modules():
            if hasattr(m, "_plans") and isinstance(m._plans, _PlanCache):
                m._plans.clear()
        return out
