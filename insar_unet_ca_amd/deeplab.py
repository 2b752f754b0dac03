"""Config 5 — DeepLabV3-CA on the same kernels: drop-in for `DeepLabV3_SingleChannel_Attn`
(/root/reference/DeepLabV3-ChannelAttention.py:83-162).

The reference builds its network out of `torchvision.models.segmentation.deeplabv3_resnet50` (:92) and re-plumbs it
(:102 classifier[4], :105-118 one-channel stem, :121 ChannelAttentionModule, :124-137 aliases, :140-162 forward).
torchvision is third-party code that is neither in /root/reference nor in the build container, so the
module tree below re-states its published structure — attribute names and registration order are what make the
state_dict interchangeable (`model.backbone.layer3.0.downsample.1.running_var`, `model.classifier.0.convs.4.1.weight`,
`aspp.project.0.weight`, `post_aspp_conv.1.bias`, `upsample_conv.weight`, `attention_module.mlp.2.weight` ...; 726
entries, 364 of them aliases) — and `forward` runs a plan of HIP launches (DeepLabPlan): every 1x1 / 3x3 / strided /
dilated convolution on the implicit-GEMM MFMA kernels, BatchNorm / ReLU / residual / pooling / dropout / bilinear resize
on the HBM-bound kernels of csrc/pointwise.hip, csrc/cam.hip and csrc/deeplab.hip. No CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib, engine, tape
from .tape import tape_py
from ._lib import InsarBnFinalize, InsarBnSeBwd, InsarCam, InsarWgrad, call, ptr
from .engine import Act, Ctx, GemmWeight, GradSink, OutConvPlan, WeightSet, _igemm, _round_up, _rows_per_part, _wgrad_nsplit
from .modules import ChannelAttentionModule, _PlanCache, _UNetFn, _require_device, _resolve_dtype

ASPP_RATES = (12, 24, 36)
GATE_STATS = os.environ.get("INSAR_GATE_STATS", "1") != "0"    # diagnostic: 0 = the last unit of a residual block reduces its (gated) incoming gradient in a pass of its own
GATE_FUSE = os.environ.get("INSAR_GATE_FUSE", "1") != "0"      # diagnostic: 0 = every residual block gates its incoming gradient in a pass of its own


# ------------------------------------------------------------------------------------------------------------
# module tree (torchvision's names)
# ------------------------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on the 3x3)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None, dilation: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class _Backbone(nn.ModuleDict):
    """IntermediateLayerGetter(resnet50(replace_stride_with_dilation=[False, True, True]), {'layer4': 'out'}): the
    ResNet's children up to layer4, under their own names (torchvision/models/_utils.py)."""

    def __init__(self):
        layers: "OrderedDict[str, nn.Module]" = OrderedDict()
        layers["conv1"] = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)      # replaced by the wrapper (:105-118)
        layers["bn1"] = nn.BatchNorm2d(64)
        layers["relu"] = nn.ReLU(inplace=True)
        layers["maxpool"] = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self_inplanes, dilation = 64, 1
        for li, (planes, blocks, stride, dilate) in enumerate(((64, 3, 1, False), (128, 4, 2, False), (256, 6, 2, True),
                                                               (512, 3, 2, True)), start=1):
            prev = dilation
            if dilate:
                dilation *= stride
                stride = 1
            ds = None
            if stride != 1 or self_inplanes != planes * 4:
                ds = nn.Sequential(nn.Conv2d(self_inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
            mods = [Bottleneck(self_inplanes, planes, stride, ds, prev)]
            self_inplanes = planes * 4
            mods += [Bottleneck(self_inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
            layers[f"layer{li}"] = nn.Sequential(*mods)
        super().__init__(layers)
        # torchvision ResNet.__init__ also owns avgpool / fc (dropped by IntermediateLayerGetter) and re-initialises:
        nn.Linear(512 * 4, 1000)                      # consumes the RNG like the dropped fc
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def _conv_bn_relu(cin, cout, k, dilation=1):
    pad = 0 if k == 1 else dilation
    return [nn.Conv2d(cin, cout, k, padding=pad, dilation=dilation, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]


class _ASPP(nn.Module):
    """torchvision deeplabv3.ASPP(2048, [12, 24, 36]): convs = [1x1, ASPPConv x3, ASPPPooling], project."""

    def __init__(self, in_channels: int = 2048, out_channels: int = 256):
        super().__init__()
        mods = [nn.Sequential(*_conv_bn_relu(in_channels, out_channels, 1))]
        mods += [nn.Sequential(*_conv_bn_relu(in_channels, out_channels, 3, r)) for r in ASPP_RATES]
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), *_conv_bn_relu(in_channels, out_channels, 1)))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(*_conv_bn_relu(len(mods) * out_channels, out_channels, 1), nn.Dropout(0.5))


class _DeepLabV3(nn.Module):
    """deeplabv3_resnet50(weights=None): backbone + DeepLabHead(2048, 21); no aux classifier."""

    def __init__(self):
        super().__init__()
        self.backbone = _Backbone()
        self.classifier = nn.Sequential(_ASPP(), nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256),
                                        nn.ReLU(), nn.Conv2d(256, 21, 1))


class DeepLabV3_SingleChannel_Attn(nn.Module):
    """Same constructor, attribute names and state_dict as the reference's wrapper (:83-137); forward (:140-162) on HIP.

    Initialisation differs from the reference's call site in ONE respect, and it matters for anyone comparing training
    curves: `deeplabv3_resnet50(pretrained=False)` (:92) still loads ImageNet weights into the BACKBONE (torchvision's
    legacy `pretrained_backbone=True` default -> `weights_backbone=IMAGENET1K_V1`, a network fetch), while this class —
    torchvision and the network being absent — starts the backbone from torch's default (kaiming) initialisation.
    To start where the reference starts, load a ResNet-50 ImageNet state_dict with `load_backbone_state_dict` (conv1 is
    mean-reduced over its three input channels exactly as :105-118 does). `pretrained=True` (the COCO-trained head, another
    fetch) is refused; a reference checkpoint made that way loads through `load_state_dict`, which drops the
    `model.aux_classifier.*` entries such a model carries (this class has no aux head: the forward pass never uses it)."""

    def __init__(self, num_classes: int = 2, backbone: str = "resnet50", pretrained: bool = False,
                 compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        if backbone != "resnet50":
            raise ValueError(f"Unsupported backbone: {backbone}" if backbone != "resnet101" else
                             "resnet101 is not part of BASELINE.json's configurations")
        if pretrained:
            raise _lib.InsarError("pretrained=True needs torchvision's downloaded weights; load a state_dict instead")
        self.model = _DeepLabV3()
        self.model.classifier[4] = nn.Conv2d(256, num_classes, kernel_size=(1, 1), stride=(1, 1))          # :102
        self.model.backbone["conv1"] = nn.Conv2d(1, 64, kernel_size=7, stride=2, padding=3, bias=False)    # :105-118
        self.attention_module = ChannelAttentionModule(in_channels=256, reduction_ratio=16)                # :121
        self.backbone = self.model.backbone                                                                 # :124
        self.aspp = self.model.classifier[0]                                                                # :126
        self.post_aspp_conv = nn.Sequential(self.model.classifier[1], self.model.classifier[2], self.model.classifier[3])
        self.upsample_conv = self.model.classifier[4]                                                       # :137
        self.compute_dtype = compute_dtype
        self._plans = _PlanCache()
        self._hooks: dict = {}

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """As nn.Module.load_state_dict, except that `model.aux_classifier.*` keys (present in checkpoints of a reference model
        built with pretrained=True: torchvision then adds an FCNHead that `forward` never calls, :140-162) are ignored."""
        aux = [k for k in state_dict if k.startswith("model.aux_classifier.")]
        if aux:
            state_dict = OrderedDict((k, v) for k, v in state_dict.items() if not k.startswith("model.aux_classifier."))
        return super().load_state_dict(state_dict, strict=strict, **kw)

    @torch.no_grad()
    def load_backbone_state_dict(self, resnet_state_dict, strict: bool = True):
        """Load a torchvision ResNet-50 state_dict (e.g. IMAGENET1K weights from a local file) into the backbone, as the
        reference's constructor effectively does (:92, `weights_backbone`): `fc.*` is dropped, and the 3-channel `conv1.weight`
        (64, 3, 7, 7) becomes the 1-channel stem by averaging over the input channels — the reference's own rule for
        pretrained stems (:105-118)."""
        sd = OrderedDict((k, v) for k, v in resnet_state_dict.items() if not k.startswith("fc."))
        w = sd.get("conv1.weight")
        if w is not None and w.shape[1] == 3:
            sd["conv1.weight"] = w.mean(dim=1, keepdim=True)
        return self.backbone.load_state_dict(sd, strict=strict)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_device(x, "DeepLabV3_SingleChannel_Attn")
        b, c, h, w = x.shape
        if c != 1:
            raise _lib.InsarError(f"DeepLabV3_SingleChannel_Attn: expected 1 input channel, got {c}")
        dt = _resolve_dtype(self)
        plan = self._plans.get((b, h, w, dt, x.device), lambda: DeepLabPlan(self, b, h, w, dt, x.device))
        for bn in plan.bn_modules:
            if bn.training != self.training:
                raise _lib.InsarError("DeepLabV3_SingleChannel_Attn: mixed BatchNorm modes are not supported by the HIP path")
        return _UNetFn.apply(plan, self.training, torch.is_grad_enabled(), self._hooks, x, *plan.grad_params)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        for m in self.modules():
            if hasattr(m, "_plans") and isinstance(m._plans, _PlanCache):
                m._plans.clear()
        return out


# ------------------------------------------------------------------------------------------------------------
# plan
# ------------------------------------------------------------------------------------------------------------
def _live(n_out: int, n_in: int, stride: int, off: int) -> bool:
    """does a tap with offset `off` touch the interior for at least one of the n_out output positions?"""
    lo = max(0, math.ceil(-off / stride))
    hi = min(n_out - 1, (n_in - 1 - off) // stride)
    return lo <= hi


class ConvUnit:
    """Conv2d(k in {1,3}, stride, dilation, bias=False) -> BatchNorm2d [-> + res] [-> ReLU]: the unit every layer of
    the backbone / ASPP / head is made of. Raw conv output `y`, activation `out` (any channel slice)."""

    def __init__(self, ctx: Ctx, conv: nn.Conv2d, bn: nn.BatchNorm2d, x: Act, out: Optional[Act], relu: bool, name: str,
                 res: Optional[Act] = None):
        self.ctx, self.conv, self.bn, self.x, self.relu, self.name, self.res = ctx, conv, bn, x, relu, name, res
        k, s, d, p = conv.kernel_size[0], conv.stride[0], conv.dilation[0], conv.padding[0]
        if conv.bias is not None or conv.groups != 1 or k not in (1, 3) or s not in (1, 2) or conv.kernel_size[1] != k:
            raise _lib.InsarError(f"{name}: unsupported convolution {conv}")
        self.k, self.s, self.d = k, s, d
        self.cin, self.cout = conv.in_channels, conv.out_channels
        if x.c_len != self.cin or self.cin % 64 or self.cout % 64:
            raise _lib.InsarError(f"{name}: channels {x.c_len} -> {self.cin}/{self.cout} must be multiples of 64")
        B, H, W = x.B, x.H, x.W
        self.Ho, self.Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
        all_taps = [(ky * d - p, kx * d - p) for ky in range(k) for kx in range(k)]
        live = [t for t, (dy, dx) in enumerate(all_taps) if _live(self.Ho, H, s, dy) and _live(self.Wo, W, s, dx)]
        # dead taps only ever multiply the zero padding: dropping them is exact. Kept simple: all taps, or the centre alone
        # (a dilated 3x3 whose rate exceeds the map, ASPP on small inputs, is a 1x1 convolution)
        self.centre_only = k == 3 and live == [4]
        self.tap_ids = [4] if self.centre_only else list(range(k * k))
        self.taps = [all_taps[t] for t in self.tap_ids]
        reach = lambda n_out, n_in, off: off < -1 or (n_out - 1) * s + off > n_in
        self.oob = any(reach(self.Ho, H, dy) or reach(self.Wo, W, dx) for dy, dx in self.taps)
        self.M = B * self.Ho * self.Wo
        self.y = Act.alloc(B, self.Ho, self.Wo, self.cout, ctx.dtype, ctx.device)
        self.out = out if out is not None else Act.alloc(B, self.Ho, self.Wo, self.cout, ctx.dtype, ctx.device)
        if self.out.c_len != self.cout or (self.out.H, self.out.W) != (self.Ho, self.Wo):
            raise _lib.InsarError(f"{name}: output slice does not match the convolution")
        # 3x3 / stride-1 convs with all nine taps, dense or dilated by <= 4 at 32 x 32 (layer1-4 and the head): the flat kernel's
        # row tiles (engine.FLAT_ROWS; conv3x3_flat.hip GEO = 1 / 2) wherever the per-tap kernel would not run 256 x 256 tiles
        self.rows_fwd = self._rows_flags(self.cout)
        self.rows_bwd = self._rows_flags(self.cin)
        self.stat_rows = (call("insar_conv3x3_flat_stat_rows", x.ref, self.cout, self.rows_fwd) if self.rows_fwd
                          else call("insar_igemm_num_mtiles", self.M, self.cout))
        self.stats = ctx.f32(self.stat_rows, 2, self.cout)
        self.stat_rps = 0 if self.stat_rows <= engine.STAT_PREFOLD_ROWS else max(64, -(-self.stat_rows // 64))
        self.fold_rows = self.stat_rows if not self.stat_rps else -(-self.stat_rows // self.stat_rps)
        self.sums = ctx.f32(self.fold_rows, 2, self.cout) if self.stat_rps else self.stats
        self.scale, self.shift, self.mean, self.invstd = (ctx.f32(self.cout) for _ in range(4))
        self.k1, self.k2 = ctx.f32(self.cout), ctx.f32(self.cout)
        self.red_rpp = _rows_per_part(B, self.Ho)
        self.red_rows = -(-self.Ho // self.red_rpp)
        self.red_part = ctx.f32(B * self.red_rows, 2, self.cout)
        self.bwd_ws = ctx.f32(B * (3 * self.cout + 1))
        self.dy: Optional[Act] = None
        self.w = GemmWeight(ctx, conv.weight, "conv3")
        self._class_w = {}
        self._tmp_grad = None
        self._ticket = None                       # last-arriver counter of the single-launch coefficient stages
        # BatchNorm-backward sums written by the epilogue of the GEMM that produces this unit's incoming gradient
        # (engine.ConvBN.bstat_slab, InsarBstat)
        self.bred, self.bred_rows, self.bred_ready = None, 0, False
        self.bred_plain, self._mask_off = False, None

    def bstat_slab(self, rows_total: int, per_image: bool, plain: bool = False):
        """(slab, (y, scale, shift)) for a GEMM that writes this unit's incoming gradient and takes its BatchNorm-backward
        sums in its epilogue. plain: sums without the unit's own ReLU mask (a residual block's last unit, whose gradient
        arrives already gated by the block's ReLU: scale 0 / shift 1 make the mask all-ones)."""
        B = self.x.B
        if (not self.relu and not plain) or (per_image and rows_total % B):
            return None
        rows = -(-rows_total // B)
        if self.bred is None or self.bred_rows != rows:
            self.bred, self.bred_rows = self.ctx.f32(B * rows, 2, self.cout), rows
        self.bred_plain = plain
        if plain:
            if self._mask_off is None:
                self._mask_off = (torch.zeros(self.cout, device=self.ctx.device), torch.ones(self.cout, device=self.ctx.device))
            return self.bred, (self.y, self._mask_off[0], self._mask_off[1])
        return self.bred, (self.y, self.scale, self.shift)

    def params(self):
        return [self.conv.weight, self.bn.weight, self.bn.bias]

    def _wptr(self, which: str) -> int:
        w = self.w.fwd() if which == "fwd" else self.w.dgrad()
        return w.data_ptr() + self.tap_ids[0] * self.cout * self.cin * self.ctx.esize

    def _rows_flags(self, N: int) -> int:
        """flip bits of the flat kernel's row tiles for this unit's convolution to N output columns (forward: cout, input
        gradient: cin), or 0: the per-tap kernel keeps 1x1 / strided / partially dead convs and its 256 x 256 tiles."""
        x = self.x
        if not engine.FLAT_ROWS or self.k != 3 or self.s != 1 or self.centre_only or len(self.taps) != 9 or x.code != _lib.BF16:
            return 0
        if self.conv.padding[0] != self.d or not call("insar_conv3x3_flat_rows_dil_ok", x.ref, N, self.d):
            return 0
        M = x.B * x.H * x.W
        if not self.oob and call("insar_igemm_tile_cols_dt", M, N, x.code) == 256:
            return 0
        narrow = (N % 128) != 0 or (M // 256) * (N // 128) < 256
        return 8 | (16 if narrow else 0) | ((self.d << 8) if self.d > 1 else 0)

    # ---- forward ------------------------------------------------------------------------------------------
    def forward(self, training: bool) -> None:
        s = _lib.stream_ptr()
        if training and self.M <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{self.x.B}, {self.cout}, {self.Ho}, {self.Wo}])")
        if self.rows_fwd:
            engine._conv3x3_flat(self.x, self.y, self.w.fwd(), 0, self.stats if training else None, geo=self.rows_fwd)
        else:
            _igemm(self.x, self.y, self._wptr("fwd"), self.cout, self.Ho, self.Wo, self.s, self.taps, 0,
                   stats=self.stats if training else None, oob=self.oob)
        if training and self.stat_rps:
            call("insar_colsum_partial", ptr(self.stats), ptr(self.sums), self.stat_rows, 2 * self.cout, self.stat_rps, s)
        bn = self.bn
        d = InsarBnFinalize()
        d.part, d.rows, d.count, d.C, d.training = ptr(self.sums), self.fold_rows, self.M, self.cout, int(training)
        d.conv_bias = 0
        d.gamma, d.beta = ptr(bn.weight), ptr(bn.bias)
        d.running_mean, d.running_var = ptr(bn.running_mean), ptr(bn.running_var)
        d.num_batches_tracked = ptr(bn.num_batches_tracked)
        d.momentum = bn.momentum if bn.momentum is not None else 0.1
        d.eps = bn.eps
        d.scale, d.shift, d.mean, d.invstd = ptr(self.scale), ptr(self.shift), ptr(self.mean), ptr(self.invstd)
        call("insar_bn_finalize", C.byref(d), s)
        if self.res is not None:
            call("insar_bn_add_relu", self.y.ref, ptr(self.scale), ptr(self.shift), self.res.ref, self.out.ref, int(self.relu), s)
        else:
            call("insar_bn_relu_apply", self.y.ref, ptr(self.scale), ptr(self.shift), 0, self.out.ref, int(self.relu), s)

    # ---- backward -----------------------------------------------------------------------------------------
    def backward(self, dout: Act, sink: GradSink, training: bool, dx: Optional[Act], relu: Optional[bool] = None,
                 add: Optional[Act] = None, bstat_for: Optional["ConvUnit"] = None, gate: Optional[Act] = None,
                 plain_for: Optional["ConvUnit"] = None) -> None:
        """dout: gradient wrt the unit's output. `relu` overrides the unit's own flag (the residual unit's caller has
        already gated dout with the block's ReLU). dx (nullable): receives the input gradient, plus `add` if given.
        bstat_for: the unit whose incoming gradient dx is (and nothing else is added to it afterwards): its BatchNorm-backward
        sums come out of the epilogue of this unit's input-gradient GEMM. gate: dx is stored as zero where this tensor is <= 0
        (the ReLU at the end of the residual block that produced this unit's input; dense per-tap GEMM only). plain_for
        (with gate): the last unit of that block — dx, as stored, is its complete incoming gradient, so its BatchNorm-backward
        sums (no mask of its own) come out of the same epilogue."""
        ctx, s = self.ctx, _lib.stream_ptr()
        relu = self.relu if relu is None else relu
        B = self.x.B
        if self.dy is None:
            self.dy = Act.alloc(B, self.Ho, self.Wo, self.cout, ctx.dtype, ctx.device)
        fused = self.bred_ready and (relu != self.bred_plain)      # the slab's sums were taken with / without this unit's ReLU mask
        self.bred_ready = False
        if not fused:
            call("insar_bnrelu_bwd_reduce", dout.ref, self.y.ref, ptr(self.scale), ptr(self.shift), ptr(self.red_part), int(relu),
                 self.red_rpp, s)
        red, red_rows = (self.bred, self.bred_rows) if fused else (self.red_part, self.red_rows)
        d = InsarBnSeBwd()
        d.B, d.H, d.W, d.C, d.Cr, d.use_se = B, self.Ho, self.Wo, self.cout, 1, 0
        d.mean, d.invstd = ptr(self.mean), ptr(self.invstd)
        d.dgamma, d.dbeta = ptr(sink.view(self.bn.weight)), ptr(sink.view(self.bn.bias))
        d.k1, d.k2 = ptr(self.k1), ptr(self.k2)
        d.accumulate = 0
        if engine.COEF_SIMPLE:      # no SE gate: one channel-parallel launch over all slab rows
            call("insar_bn_bwd_coef", C.byref(d), ptr(red), red_rows * B, ptr(self.scale), 0, int(training), s)
        elif engine.COEF_FUSE:      # both coefficient stages in one launch (stage 2 by the last-arriving work-group)
            if self._ticket is None:
                self._ticket = torch.zeros(1, dtype=torch.int32, device=ctx.device)
            call("insar_bnse_bwd_coef_fused", C.byref(d), ptr(red), red_rows, ptr(self.scale), ptr(self.shift),
                 ptr(self.bwd_ws), 0, int(training), ptr(self._ticket), s)
        else:
            call("insar_bnse_bwd_coef", C.byref(d), ptr(red), red_rows, ptr(self.scale), ptr(self.shift),
                 ptr(self.bwd_ws), 0, int(training), s)
        call("insar_bnrelu_bwd_apply", dout.ref, self.y.ref, ptr(self.scale), ptr(self.shift), ptr(self.mean), ptr(self.invstd),
             0, 0, ptr(self.k1), ptr(self.k2), self.dy.ref, int(relu), s)
        with ctx.side_stream():
            self._weight_grad(sink.view(self.conv.weight))
        if dx is not None:
            self._input_grad(dx, add, bstat_for, gate, plain_for)

    def can_gate(self) -> bool:
        """The input-gradient GEMM of this unit can apply a ReLU mask to what it stores (stride 1, per-tap GEMM)."""
        return self.s == 1 and not self.rows_bwd

    def _input_grad(self, dx: Act, add: Optional[Act], bstat_for: Optional["ConvUnit"] = None, gate: Optional[Act] = None,
                    plain_for: Optional["ConvUnit"] = None) -> None:
        H, W = self.x.H, self.x.W
        if gate is not None and not self.can_gate():
            raise _lib.InsarError(f"{self.name}: no gated input gradient on this path")
        if self.s == 1:
            taps = [(-dy, -dx_) for dy, dx_ in self.taps]
            slab = None
            if self.rows_bwd and add is None:
                # (the dilated 128-column instantiation with the sums in its epilogue spills: 130 us against 68 + a 20 us reduce pass)
                heavy = (self.rows_bwd >> 8) and not (self.rows_bwd & 16)
                if bstat_for is not None and engine.BSTAT_FUSE and not heavy and engine._same_layout(dx, bstat_for.y):
                    slab = bstat_for.bstat_slab(call("insar_conv3x3_flat_stat_rows", self.dy.ref, self.cin, self.rows_bwd), False)
                engine._conv3x3_flat(self.dy, dx, self.w.dgrad(), 1, slab[0] if slab else None, bstat=slab[1] if slab else None,
                                     geo=self.rows_bwd)
                if slab:
                    bstat_for.bred_ready = True
                return
            if add is None and gate is None and bstat_for is not None:
                slab = engine._igemm_bstat_slab(bstat_for, False, self.x.B * H * W, self.cin, H * W, dx)
            elif gate is not None and plain_for is not None and engine.BSTAT_FUSE and GATE_STATS and engine._same_layout(dx, plain_for.y):
                slab = plain_for.bstat_slab(call("insar_igemm_num_mtiles", self.x.B * H * W, self.cin), False, plain=True)
                bstat_for = plain_for
            _igemm(self.dy, dx, self._wptr("dgrad"), self.cin, H, W, 1, taps, 0, oob=self.oob, add=add,
                   stats=slab[0] if slab else None, bstat=slab[1] if slab else None, gate=gate)
            if slab:
                bstat_for.bred_ready = True
            return
        # stride 2: the input gradient of a strided convolution, one launch per parity class of the input pixel
        wd = self.w.dgrad()                                     # [T][Ci][Co]
        for py in (0, 1):
            for px in (0, 1):
                sel = [(i, t) for i, t in enumerate(self.taps) if (t[0] - py) % 2 == 0 and (t[1] - px) % 2 == 0]
                hc, wc = (H - py + 1) // 2, (W - px + 1) // 2
                if not sel or hc <= 0 or wc <= 0:
                    if add is None and hc > 0 and wc > 0:
                        raise _lib.InsarError(f"{self.name}: a parity class of the input gradient has no tap and nothing to add to")
                    continue
                key = (py, px)
                if key not in self._class_w:
                    idx = torch.tensor([self.tap_ids[i] for i, _ in sel], dtype=torch.int64, device=self.ctx.device)
                    self._class_w[key] = (idx, torch.empty((len(sel),) + tuple(wd.shape[1:]), dtype=wd.dtype, device=wd.device))
                idx, buf = self._class_w[key]
                tape_py(lambda wd=wd, idx=idx, buf=buf: torch.index_select(wd, 0, idx, out=buf))
                taps = [((py - t[0]) // 2, (px - t[1]) // 2) for _, t in sel]
                _igemm(self.dy, dx, buf, self.cin, hc, wc, 1, taps, 0, add=add, out_stride=2, out_off=(py, px))

    def _weight_grad(self, gw: torch.Tensor) -> None:
        ctx = self.ctx
        x, dy = self.x, self.dy
        if self.k == 3 and self.s == 1 and self.d == 1 and not self.centre_only:
            engine._wgrad_conv3(ctx, x, dy, gw)
            return
        B = x.B
        nt = len(self.taps)
        tabx = ctx.taps_table(B, self.Ho, self.Wo, self.s, x.H, x.W, self.taps)
        tabdy = ctx.pixel_table(B, self.Ho, self.Wo, 1, self.Ho, self.Wo, 0)
        mpad = tabdy.numel()
        tm, tn = engine._wgrad_tiles(self.cin, self.cout, ctx.code)
        tiles = nt * (self.cin // tm) * (self.cout // tn)
        alone = ctx.side is None or (engine.PROFILER is not None and engine.PROFILER.alone)
        fill = 1.0 if alone else engine._side_fill(ctx, engine.WGRAD_FILL_DL)
        nsplit = _wgrad_nsplit(tiles, mpad // engine.WG_BKP, nt * self.cout * self.cin, tm, tn, ctx.esize, fill=fill)
        part = ctx.wgrad_part(nsplit * nt * self.cout * self.cin)
        d = InsarWgrad()
        d.x, d.dy = x.desc, dy.desc
        d.tabx, d.tabdy, d.part = ptr(tabx), ptr(tabdy), ptr(part)
        d.Mpad, d.nsplit, d.ntaps = mpad, nsplit, nt
        d.tabx_tap_stride = mpad
        engine._launch_wgrad(d, self.M, self.cin, self.cout, nt, ctx.code)
        if self.centre_only:
            if self._tmp_grad is None:
                self._tmp_grad = ctx.f32(self.cout, self.cin)
            ctx.wgrad_finish(part, self._tmp_grad, nsplit, 1, self.cout, self.cin, 0)
            centre, tmp = gw[:, :, 1, 1], self._tmp_grad
            tape_py(lambda: (gw.zero_(), centre.copy_(tmp)))
        else:
            ctx.wgrad_finish(part, gw, nsplit, nt, self.cout, self.cin, 0)


class BottleneckPlan:
    def __init__(self, ctx: Ctx, mod: Bottleneck, x: Act, name: str):
        self.ctx, self.mod, self.x, self.name = ctx, mod, x, name
        self.u1 = ConvUnit(ctx, mod.conv1, mod.bn1, x, None, True, name + ".conv1")
        self.u2 = ConvUnit(ctx, mod.conv2, mod.bn2, self.u1.out, None, True, name + ".conv2")
        self.ud = None
        res = x
        if mod.downsample is not None:
            self.ud = ConvUnit(ctx, mod.downsample[0], mod.downsample[1], x, None, False, name + ".downsample")
            res = self.ud.out
        self.u3 = ConvUnit(ctx, mod.conv3, mod.bn3, self.u2.out, None, True, name + ".conv3", res=res)
        self.out = self.u3.out
        self.dout: Optional[Act] = None          # gradient wrt the block output (written by the consumer)
        self.dout_gated = False                  # the consumer has already applied this block's ReLU mask to dout (see backward)
        self.dz1 = self.dz2 = None

    def units(self):
        return [self.u1, self.u2, self.u3] + ([self.ud] if self.ud else [])

    def params(self):
        ps = self.u1.params() + self.u2.params() + self.u3.params()
        return ps + (self.ud.params() if self.ud else [])

    def forward(self, training: bool) -> None:
        self.u1.forward(training)
        self.u2.forward(training)
        if self.ud is not None:
            self.ud.forward(training)
        self.u3.forward(training)

    def grad_out(self) -> Act:
        if self.dout is None:
            o = self.out
            self.dout = Act.alloc(o.B, o.H, o.W, o.c_len, self.ctx.dtype, self.ctx.device)
        return self.dout

    def backward(self, sink: GradSink, training: bool, dx: Act, producer: Optional["BottleneckPlan"] = None) -> None:
        """dx: gradient wrt the block input. producer: the residual block whose output IS this block's input (dx is its
        `dout`): the GEMM that writes dx last applies that block's ReLU mask (x > 0) to what it stores, and the producer
        skips its own pass over dout (16 x 3 passes over the largest activations of the network, config 5: 0.38 ms)."""
        ctx = self.ctx
        g = self.grad_out()
        if self.dz1 is None:
            a, b = self.u1.out, self.u2.out
            self.dz1 = Act.alloc(a.B, a.H, a.W, a.c_len, ctx.dtype, ctx.device)
            self.dz2 = Act.alloc(b.B, b.H, b.W, b.c_len, ctx.dtype, ctx.device)
        # out = relu(bn3(conv3) + identity): gate the incoming gradient once, in place; both branches take it
        if not self.dout_gated:
            call("insar_relu_gate_bwd", g.ref, self.out.ref, g.ref, _lib.stream_ptr())
        self.dout_gated = False
        last = self.u1 if self.ud is None else self.ud
        gate = self.x if (GATE_FUSE and producer is not None and producer.out is self.x and last.can_gate()) else None
        self.u3.backward(g, sink, training, self.dz2, relu=False, bstat_for=self.u2)
        self.u2.backward(self.dz2, sink, training, self.dz1, bstat_for=self.u1)
        if self.ud is None:
            self.u1.backward(self.dz1, sink, training, dx, add=g, gate=gate,            # identity branch: dx = dgrad + g
                             plain_for=producer.u3 if gate is not None else None)
        else:
            self.u1.backward(self.dz1, sink, training, dx)
            self.ud.backward(g, sink, training, dx, relu=False, add=dx, gate=gate, plain_for=producer.u3 if gate is not None else None)
        if gate is not None:
            producer.dout_gated = True


class DeepLabPlan(tape.PlanTape):
    """Buffers + launch sequence of DeepLabV3_SingleChannel_Attn.forward / backward for one input geometry."""

    def __init__(self, net: DeepLabV3_SingleChannel_Attn, B: int, H: int, W: int, dtype: torch.dtype, device: torch.device):
        if H % 8 or W % 8:
            raise _lib.InsarError(f"H={H}, W={W}: the HIP path of DeepLabV3-CA covers inputs that are multiples of 8 (output stride 8)")
        self.net, self.B, self.H, self.W = net, B, H, W
        ctx = self.ctx = Ctx(device, dtype)
        A = lambda h, w, c: Act.alloc(B, h, w, c, dtype, device)
        bb, head = net.model.backbone, net.model.classifier
        # stem
        self.stem_conv, self.stem_bn = bb["conv1"], bb["bn1"]
        h2, w2, h4, w4 = H // 2, W // 2, H // 4, W // 4
        self.y0, self.z0, self.p0 = A(h2, w2, 64), A(h2, w2, 64), A(h4, w4, 64)
        self.pool_arg = torch.zeros((B, h4, w4, 64), dtype=torch.uint8, device=device)
        self.st_rows = call("insar_conv7x7s2_fwd_rows", B, H)
        self.st_stats = ctx.f32(self.st_rows, 2, 64)
        self.st_rps = 0 if self.st_rows <= 256 else max(64, -(-self.st_rows // 64))
        self.st_fold = self.st_rows if not self.st_rps else -(-self.st_rows // self.st_rps)
        self.st_sums = ctx.f32(self.st_fold, 2, 64) if self.st_rps else self.st_stats
        self.st_scale, self.st_shift, self.st_mean, self.st_invstd, self.st_k1, self.st_k2 = (ctx.f32(64) for _ in range(6))
        self.st_rpp = _rows_per_part(B, h2)
        self.st_red_rows = -(-h2 // self.st_rpp)
        self.st_red = ctx.f32(B * self.st_red_rows, 2, 64)
        self.st_ws = ctx.f32(B * (3 * 64 + 1))
        self.st_nb = call("insar_conv7x7s2_wgrad_blocks", B, h2)
        self.st_part = ctx.f32(self.st_nb, 64 * 49)
        self.dz0 = self.dy0 = self.dp0 = None
        # residual layers
        self.blocks: List[BottleneckPlan] = []
        self.layer_blocks: List[List[BottleneckPlan]] = []
        x = self.p0
        for li in range(1, 5):
            grp = []
            for bi, mod in enumerate(bb[f"layer{li}"]):
                blk = BottleneckPlan(ctx, mod, x, f"layer{li}.{bi}")
                grp.append(blk)
                self.blocks.append(blk)
                x = blk.out
            self.layer_blocks.append(grp)
        self.x5 = x
        h8, w8 = x.H, x.W
        # ASPP
        aspp = head[0]
        self.cat = A(h8, w8, 1280)
        self.branches = [ConvUnit(ctx, aspp.convs[i][0], aspp.convs[i][1], self.x5, self.cat.slice(256 * i, 256), True, f"aspp.convs.{i}")
                         for i in range(4)]
        # The pooling branch runs in fp32 whatever the compute type (B x 2048 values): the pooled vectors of similar tiles
        # differ by far less than bf16 resolves, and the BatchNorm behind the 1x1 conv divides by that spread.
        self.ctx32 = ctx
        if dtype != torch.float32:
            self.ctx32 = Ctx(device, torch.float32)
            self.ctx32.side = None               # its (tiny) weight gradient stays on the main stream
        self.gp = Act.alloc(B, 1, 1, 2048, torch.float32, device)
        self.pool_unit = ConvUnit(self.ctx32, aspp.convs[4][1], aspp.convs[4][2], self.gp, None, True, "aspp.convs.4")
        self.project = ConvUnit(ctx, aspp.project[0], aspp.project[1], self.cat, None, True, "aspp.project")
        self.drop_p = float(aspp.project[3].p)
        self.zdrop = A(h8, w8, 256)
        self.drop_mask = torch.zeros((B, h8, w8, 256), dtype=torch.uint8, device=device)
        self.drop_active = False
        self.drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.drop_counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.external_mask = False        # tests: apply a caller-supplied mask instead of drawing one
        # head: 3x3 conv + BN + ReLU (post_aspp_conv), ChannelAttentionModule, 1x1 classifier, bilinear resize
        self.head = ConvUnit(ctx, head[1], head[2], self.zdrop, None, True, "classifier.1")
        cam = net.attention_module
        self.cam_mod, self.cam_cr = cam, cam.mlp[0].out_channels
        self.zc = A(h8, w8, 256)
        self.cam_rpp = _rows_per_part(B, h8)
        self.cam_rows = -(-h8 // self.cam_rpp)
        f = ctx.f32
        self.cam_psum, self.cam_pmax = f(B * self.cam_rows, 256), f(B * self.cam_rows, 256)
        self.cam_parg = torch.zeros(B * self.cam_rows, 256, dtype=torch.int32, device=device)
        self.cam_avg, self.cam_mx, self.cam_gate = f(B, 256), f(B, 256), f(B, 256)
        self.cam_arg = torch.zeros(B, 256, dtype=torch.int32, device=device)
        self.cam_ha, self.cam_hm = f(B, self.cam_cr), f(B, self.cam_cr)
        self.cam_coefB, self.cam_dmax = f(B, 256), f(B, 256)
        self.cam_red = f(B * self.cam_rows, 2, 256)
        self.cam_ws = f(B * (256 + 2 * self.cam_cr))
        self.ones, self.zeros = ctx.const(1.0, 256), ctx.const(0.0, 256)
        self.outc = OutConvPlan(ctx, head[4], self.zc)
        self.K = head[4].out_channels
        self.logits_lo = None
        # gradient buffers of the head side (allocated on first backward)
        self._g = {}
        # parameters grouped by backward stage (completion order), for the flat gradient buffer / DP buckets
        cam_params = [cam.mlp[0].weight, cam.mlp[2].weight]
        groups = [[head[4].weight, head[4].bias] + cam_params + self.head.params(),
                  self.project.params() + self.pool_unit.params() + [p for u in self.branches for p in u.params()]]
        for grp in reversed(self.layer_blocks):
            groups.append([p for blk in reversed(grp) for p in blk.params()])
        groups[-1] = groups[-1] + [self.stem_conv.weight, self.stem_bn.weight, self.stem_bn.bias]
        self.grad_params = [p for g in groups for p in g]
        if len({id(p) for p in self.grad_params}) != len(list(net.parameters())):
            raise _lib.InsarError("DeepLabPlan: the gradient layout does not cover every parameter")
        self.sink = GradSink(ctx, None, groups)
        self.stage_sizes = self.sink.group_sizes
        self.stage_ends = [sum(self.stage_sizes[:i + 1]) for i in range(len(self.stage_sizes))]
        self._closes = {}
        units = [u for blk in self.blocks for u in blk.units()] + self.branches + [self.pool_unit, self.project, self.head]
        self.units = units
        self.weightset = WeightSet(ctx, [u.w for u in units])
        self.bn_modules = [self.stem_bn] + [u.bn for u in units]
        self._logits_lo = self._dlo = None
        self._tape_setup()
        self.busy = False
        self.training = True
        self.x_in: Optional[torch.Tensor] = None

    def bucket_closes(self, min_elems: int):
        if min_elems not in self._closes:
            from .parallel import plan_buckets
            self._closes[min_elems] = set(plan_buckets(self.stage_sizes, min_elems))
        return self._closes[min_elems]

    def _grad(self, key: str, like: Act) -> Act:
        if key not in self._g:
            self._g[key] = Act.alloc(like.B, like.H, like.W, like.c_len, like.buf.dtype, self.ctx.device)
        return self._g[key]

    def _cam_desc(self) -> InsarCam:
        d = InsarCam()
        z = self.head.out
        d.B, d.H, d.W, d.C, d.Cr, d.rows = z.B, z.H, z.W, 256, self.cam_cr, self.cam_rows
        d.psum, d.pmax, d.parg = ptr(self.cam_psum), ptr(self.cam_pmax), ptr(self.cam_parg)
        d.w1, d.w2 = ptr(self.cam_mod.mlp[0].weight), ptr(self.cam_mod.mlp[2].weight)
        d.avg, d.mx, d.arg = ptr(self.cam_avg), ptr(self.cam_mx), ptr(self.cam_arg)
        d.ha, d.hm, d.gate = ptr(self.cam_ha), ptr(self.cam_hm), ptr(self.cam_gate)
        d.coefB, d.dmax, d.ws = ptr(self.cam_coefB), ptr(self.cam_dmax), ptr(self.cam_ws)
        d.accumulate = 0
        return d

    # ---- forward (DeepLabV3-ChannelAttention.py:140-162) -------------------------------------------------------
    def _tape_key(self, which: str) -> tuple:
        return super()._tape_key(which) + (self.drop_p,)

    def forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        """The ordinary launch sequence (_forward_eager) or, in the steady state of a training loop, its launch tape (tape.py)."""
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        if not self._tape_allowed(training, not self.external_mask):
            return self._forward_eager(x, training)
        logits = torch.empty((self.B, self.K, self.H, self.W), dtype=torch.float32, device=self.ctx.device)
        out, replayed = self._run(self._tape_key("f"), lambda: self._forward_eager(x, training),
                                  {"x": x.data_ptr(), "logits": logits.data_ptr()}, {x.data_ptr(): "x"},
                                  dyn_after=lambda o: {o.data_ptr(): "logits"})
        if replayed:
            self.training = training
            self.x_in = x.detach()                 # the stem's weight gradient reads it in backward
            self.drop_active = training and self.drop_p > 0.0
            return logits
        return out

    def _forward_eager(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        s = _lib.stream_ptr()
        ctx = self.ctx
        self.training = training
        self.x_in = x.detach()
        with ctx.side_stream():
            self.weightset.refresh()
        # stem (:144 backbone): conv7x7 s2 -> BN -> ReLU -> MaxPool(3, 2, 1)
        if training and self.B * (self.H // 2) * (self.W // 2) <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        call("insar_conv7x7s2_fwd", ptr(self.x_in), self.H, self.W, ptr(self.stem_conv.weight), self.y0.ref,
             ptr(self.st_stats) if training else 0, s)
        if training and self.st_rps:
            call("insar_colsum_partial", ptr(self.st_stats), ptr(self.st_sums), self.st_rows, 128, self.st_rps, s)
        bn = self.stem_bn
        d = InsarBnFinalize()
        d.part, d.rows, d.count, d.C, d.training = ptr(self.st_sums), self.st_fold, self.B * (self.H // 2) * (self.W // 2), 64, int(training)
        d.conv_bias = 0
        d.gamma, d.beta = ptr(bn.weight), ptr(bn.bias)
        d.running_mean, d.running_var, d.num_batches_tracked = ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked)
        d.momentum, d.eps = (bn.momentum if bn.momentum is not None else 0.1), bn.eps
        d.scale, d.shift, d.mean, d.invstd = ptr(self.st_scale), ptr(self.st_shift), ptr(self.st_mean), ptr(self.st_invstd)
        call("insar_bn_finalize", C.byref(d), s)
        call("insar_bn_relu_apply", self.y0.ref, ptr(self.st_scale), ptr(self.st_shift), 0, self.z0.ref, 1, s)
        call("insar_maxpool3s2_fwd", self.z0.ref, self.p0.ref, ptr(self.pool_arg), s)
        ctx.join_side()                       # GEMM-layout weights are ready
        for blk in self.blocks:
            blk.forward(training)
        # ASPP (:148)
        for u in self.branches:
            u.forward(training)
        hw = self.x5.H * self.x5.W
        call("insar_sum_hw", self.x5.ref, self.gp.ref, 1.0 / hw, s)
        self.pool_unit.forward(training)
        call("insar_broadcast_hw", self.pool_unit.out.ref, self.cat.slice(1024, 256).ref, 1.0, 0, s)
        self.project.forward(training)
        self.drop_active = training and self.drop_p > 0.0
        if self.drop_active:
            # mask = hash(seed drawn once from torch's RNG, device-side forward counter, element index): a new mask every
            # training forward, also when the step is replayed from a captured hipGraph
            ctr = self.drop_counter
            tape_py(lambda: ctr.add_(1))
            call("insar_dropout", self.project.out.ref, self.zdrop.ref, ptr(self.drop_mask), self.drop_seed, ptr(self.drop_counter),
                 self.drop_p, 0 if self.external_mask else 1, s)
            self.head.x = self.zdrop
        else:
            self.head.x = self.project.out
        # head (:151 post_aspp_conv, :154 attention, :157 upsample_conv, :160 resize)
        self.head.forward(training)
        z = self.head.out
        call("insar_cam_pool", z.ref, ptr(self.cam_psum), ptr(self.cam_pmax), ptr(self.cam_parg), self.cam_rpp, s)
        dcam = self._cam_desc()
        call("insar_cam_excite", C.byref(dcam), s)
        call("insar_bn_relu_apply", z.ref, ptr(self.ones), ptr(self.zeros), ptr(self.cam_gate), self.zc.ref, 0, s)
        if self._logits_lo is None:
            self._logits_lo = torch.empty((self.B, self.K, z.H, z.W), dtype=torch.float32, device=ctx.device)
        self.logits_lo = self.outc.forward(out=self._logits_lo)       # plan-owned (a launch tape holds its address)
        logits = torch.empty((self.B, self.K, self.H, self.W), dtype=torch.float32, device=ctx.device)
        call("insar_bilinear_fwd", ptr(self.logits_lo), ptr(logits), self.B * self.K, z.H, z.W, self.H, self.W, s)
        return logits

    # ---- backward ------------------------------------------------------------------------------------------------
    def backward(self, dlogits: torch.Tensor, on_bucket=None) -> List[torch.Tensor]:
        """The ordinary launch sequence (_backward_eager) or its launch tape (tape.py)."""
        if dlogits.dtype != torch.float32 or not dlogits.is_contiguous():
            dlogits = dlogits.float().contiguous()
        self.sink.select()
        if not self._tape_allowed(self.training, on_bucket is None and not self.external_mask and not self.weightset.stale()):
            return self._backward_eager(dlogits, on_bucket)
        xp = self.x_in.data_ptr()
        out, replayed = self._run(self._tape_key("b"), lambda: self._backward_eager(dlogits, None),
                                  {"dlogits": dlogits.data_ptr(), "x": xp}, {dlogits.data_ptr(): "dlogits", xp: "x"})
        if replayed:
            return [self.sink.view(p) for p in self.grad_params]
        return out

    def _backward_eager(self, dlogits: torch.Tensor, on_bucket=None) -> List[torch.Tensor]:
        s = _lib.stream_ptr()
        ctx, sink, training = self.ctx, self.sink, self.training
        z = self.head.out
        if self._dlo is None:
            self._dlo = torch.empty_like(self.logits_lo)
        dlo = self._dlo
        call("insar_bilinear_bwd", ptr(dlogits), ptr(dlo), self.B * self.K, z.H, z.W, self.H, self.W, s)
        dzc = self._grad("dzc", self.zc)
        self.outc.backward(dlo, sink, dzc)
        # ChannelAttentionModule backward (csrc/cam.hip)
        dz = self._grad("dz", z)
        call("insar_bnrelu_bwd_reduce", dzc.ref, z.ref, ptr(self.ones), ptr(self.zeros), ptr(self.cam_red), 0, self.cam_rpp, s)
        dcam = self._cam_desc()
        w1, w2 = self.cam_mod.mlp[0].weight, self.cam_mod.mlp[2].weight
        dcam.dw1, dcam.dw2 = ptr(sink.view(w1)), ptr(sink.view(w2))
        call("insar_cam_bwd_coef", C.byref(dcam), ptr(self.cam_red), self.cam_rows, s)
        call("insar_bnrelu_bwd_apply", dzc.ref, z.ref, ptr(self.ones), ptr(self.zeros), ptr(self.zeros), ptr(self.ones),
             ptr(self.cam_gate), ptr(self.cam_coefB), ptr(self.zeros), ptr(self.zeros), dz.ref, 0, s)
        call("insar_cam_scatter_max", dz.ref, ptr(self.cam_dmax), ptr(self.cam_arg), s)
        dzdrop = self._grad("dzdrop", self.zdrop)
        self.head.backward(dz, sink, training, dzdrop)
        if on_bucket is not None:
            on_bucket(self, ("head", 0))
        if self.drop_active:
            dproj = self._grad("dproj", self.project.out)
            call("insar_dropout", dzdrop.ref, dproj.ref, ptr(self.drop_mask), 0, 0, self.drop_p, 0, s)
        else:
            dproj = dzdrop
        dcat = self._grad("dcat", self.cat)
        self.project.backward(dproj, sink, training, dcat)
        dx5 = self.blocks[-1].grad_out()
        dzp = self._grad("dzp", self.pool_unit.out)
        call("insar_sum_hw", dcat.slice(1024, 256).ref, dzp.ref, 1.0, s)
        dgp = self._grad("dgp", self.gp)
        self.pool_unit.backward(dzp, sink, training, dgp)
        for i, u in enumerate(self.branches):
            u.backward(dcat.slice(256 * i, 256), sink, training, dx5, add=dx5 if i > 0 else None)
        last = self.blocks[-1]
        if GATE_FUSE and last.out is self.x5:      # the last writer of layer4's incoming gradient applies the last block's ReLU mask
            call("insar_broadcast_hw_gate", dgp.ref, dx5.ref, last.out.ref, 1.0 / (self.x5.H * self.x5.W), 1, s)
            last.dout_gated = True
        else:
            call("insar_broadcast_hw", dgp.ref, dx5.ref, 1.0 / (self.x5.H * self.x5.W), 1, s)
        if on_bucket is not None:
            on_bucket(self, ("aspp", 0))
        # residual layers, last to first
        if self.dp0 is None:
            self.dp0 = Act.alloc(self.p0.B, self.p0.H, self.p0.W, 64, ctx.dtype, ctx.device)
            self.dz0 = Act.alloc(self.z0.B, self.z0.H, self.z0.W, 64, ctx.dtype, ctx.device)
            self.dy0 = Act.alloc(self.z0.B, self.z0.H, self.z0.W, 64, ctx.dtype, ctx.device)
        for li in (3, 2, 1, 0):
            grp = self.layer_blocks[li]
            for bi in range(len(grp) - 1, -1, -1):
                blk = grp[bi]
                prev = grp[bi - 1] if bi > 0 else (self.layer_blocks[li - 1][-1] if li > 0 else None)
                dx = prev.grad_out() if prev is not None else self.dp0
                blk.backward(sink, training, dx, prev)
            if li > 0 and on_bucket is not None:
                on_bucket(self, ("layer", li + 1))
        # stem backward: MaxPool gradient, BN + ReLU backward, weight gradient of the 7x7 conv
        call("insar_maxpool3s2_bwd", self.dp0.ref, ptr(self.pool_arg), self.dz0.ref, s)
        call("insar_bnrelu_bwd_reduce", self.dz0.ref, self.y0.ref, ptr(self.st_scale), ptr(self.st_shift), ptr(self.st_red), 1,
             self.st_rpp, s)
        d = InsarBnSeBwd()
        d.B, d.H, d.W, d.C, d.Cr, d.use_se = self.B, self.z0.H, self.z0.W, 64, 1, 0
        d.mean, d.invstd = ptr(self.st_mean), ptr(self.st_invstd)
        d.dgamma, d.dbeta = ptr(sink.view(self.stem_bn.weight)), ptr(sink.view(self.stem_bn.bias))
        d.k1, d.k2 = ptr(self.st_k1), ptr(self.st_k2)
        d.accumulate = 0
        call("insar_bnse_bwd_coef", C.byref(d), ptr(self.st_red), self.st_red_rows, ptr(self.st_scale), ptr(self.st_shift),
             ptr(self.st_ws), 0, int(training), s)
        call("insar_bnrelu_bwd_apply", self.dz0.ref, self.y0.ref, ptr(self.st_scale), ptr(self.st_shift), ptr(self.st_mean),
             ptr(self.st_invstd), 0, 0, ptr(self.st_k1), ptr(self.st_k2), self.dy0.ref, 1, s)
        with ctx.side_stream():
            call("insar_conv7x7s2_wgrad", ptr(self.x_in), self.H, self.W, self.dy0.ref, ptr(self.st_part), _lib.stream_ptr())
            ctx.colsum(self.st_part, sink.view(self.stem_conv.weight).view(-1), 1, self.st_nb, 64 * 49)
        if on_bucket is not None:
            on_bucket(self, ("layer", 1))
        ctx.join_side()
        return [sink.view(p) for p in self.grad_params]
