"""TEST INFRASTRUCTURE ONLY — CPU restatement of config 5, `DeepLabV3_SingleChannel_Attn`
(/root/reference/DeepLabV3-ChannelAttention.py:83-162).

PARITY UNPINNED for everything but `ChannelAttentionModule`: the arithmetic of the backbone, the ASPP and
the head lives in torchvision (`torchvision.models.segmentation.deeplabv3_resnet50`, `torchvision.models.resnet`),
which is absent from the build container and from /root/reference, whose version is not pinned by the reference (no
requirements file), and whose constructor fetches ImageNet weights from the network. This file restates the PUBLISHED
architecture of torchvision (resnet.py: ResNet / Bottleneck v1.5 with replace_stride_with_dilation=[False, True, True];
segmentation/deeplabv3.py: DeepLabHead / ASPP / ASPPConv / ASPPPooling; segmentation/_utils.py: IntermediateLayerGetter
{layer4 -> "out"}, no aux classifier when weights is None) as the reference's own call sites use it:
  :92        deeplabv3_resnet50(pretrained=False)             -> backbone + classifier, 21 classes
  :102       classifier[4] = Conv2d(256, num_classes, 1)      (with bias)
  :105-118   backbone.conv1 = Conv2d(1, 64, 7, stride 2, padding 3, bias=False)
  :121       attention_module = ChannelAttentionModule(256, 16)   (pinned by fixture G7, oracle/unet_ca_oracle.cam_layer)
  :124-137   backbone / aspp / post_aspp_conv (= classifier[1..3]: the 3x3 conv, BN, ReLU) / upsample_conv (= classifier[4])
             are ALIASES of sub-modules of `model`, so the state_dict lists those tensors under both names
  :140-162   forward: backbone['out'] -> aspp -> post_aspp_conv -> attention -> upsample_conv -> F_T.resize(BILINEAR)
The restatement is functional and state_dict driven, like oracle/unet_ca_oracle.py; gradients come from autograd.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn.functional as F

from .unet_ca_oracle import cam_layer, cross_entropy, is_param  # noqa: F401  (same CE / CAM arithmetic)

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
LAYERS = (3, 4, 6, 3)                     # resnet50
PLANES = (64, 128, 256, 512)
ASPP_RATES = (12, 24, 36)
DROPOUT_P = 0.5


def block_specs():
    """(prefix, inplanes, planes, stride, dilation, has_downsample) of every Bottleneck, torchvision's _make_layer with
    replace_stride_with_dilation = [False, True, True]: layer3 / layer4 trade their stride for dilation; the first block
    of a layer runs at the PREVIOUS dilation, the others at the new one."""
    specs, inplanes, dilation = [], 64, 1
    for li, (nblocks, planes) in enumerate(zip(LAYERS, PLANES)):
        stride = 1 if li == 0 else 2
        dilate = li >= 2
        prev = dilation
        if dilate:
            dilation *= stride
            stride = 1
        for b in range(nblocks):
            first = b == 0
            specs.append((f"layer{li + 1}.{b}", inplanes if first else planes * 4, planes, stride if first else 1,
                          prev if first else dilation, first and (stride != 1 or inplanes != planes * 4)))
            if first:
                inplanes = planes * 4
    return specs


def _bn_entries(prefix, c):
    return [(f"{prefix}.weight", (c,)), (f"{prefix}.bias", (c,)), (f"{prefix}.running_mean", (c,)),
            (f"{prefix}.running_var", (c,)), (f"{prefix}.num_batches_tracked", ())]


def _backbone_entries(p):
    e = [(f"{p}.conv1.weight", (64, 1, 7, 7))] + _bn_entries(f"{p}.bn1", 64)
    for name, cin, planes, _s, _d, ds in block_specs():
        q = f"{p}.{name}"
        e.append((f"{q}.conv1.weight", (planes, cin, 1, 1))); e += _bn_entries(f"{q}.bn1", planes)
        e.append((f"{q}.conv2.weight", (planes, planes, 3, 3))); e += _bn_entries(f"{q}.bn2", planes)
        e.append((f"{q}.conv3.weight", (planes * 4, planes, 1, 1))); e += _bn_entries(f"{q}.bn3", planes * 4)
        if ds:
            e.append((f"{q}.downsample.0.weight", (planes * 4, cin, 1, 1))); e += _bn_entries(f"{q}.downsample.1", planes * 4)
    return e


def _aspp_entries(p):
    e = [(f"{p}.convs.0.0.weight", (256, 2048, 1, 1))] + _bn_entries(f"{p}.convs.0.1", 256)
    for i in (1, 2, 3):
        e.append((f"{p}.convs.{i}.0.weight", (256, 2048, 3, 3))); e += _bn_entries(f"{p}.convs.{i}.1", 256)
    e.append((f"{p}.convs.4.1.weight", (256, 2048, 1, 1))); e += _bn_entries(f"{p}.convs.4.2", 256)
    e.append((f"{p}.project.0.weight", (256, 1280, 1, 1))); e += _bn_entries(f"{p}.project.1", 256)
    return e


def state_dict_template(num_classes: int = 2) -> "OrderedDict[str, torch.Tensor]":
    """Names, shapes and ORDER of `DeepLabV3_SingleChannel_Attn(...).state_dict()`; aliased entries share storage."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def add(entries):
        for k, shp in entries:
            sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)

    add(_backbone_entries("model.backbone"))
    add(_aspp_entries("model.classifier.0"))
    add([("model.classifier.1.weight", (256, 256, 3, 3))] + _bn_entries("model.classifier.2", 256))
    add([("model.classifier.4.weight", (num_classes, 256, 1, 1)), ("model.classifier.4.bias", (num_classes,))])
    add([("attention_module.mlp.0.weight", (16, 256, 1, 1)), ("attention_module.mlp.2.weight", (256, 16, 1, 1))])
    for alias, src in (("backbone", "model.backbone"), ("aspp", "model.classifier.0")):
        for k in [k for k in sd if k.startswith(src + ".")]:
            sd[alias + k[len(src):]] = sd[k]
    sd["post_aspp_conv.0.weight"] = sd["model.classifier.1.weight"]
    for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
        sd[f"post_aspp_conv.1.{leaf}"] = sd[f"model.classifier.2.{leaf}"]
    sd["upsample_conv.weight"] = sd["model.classifier.4.weight"]
    sd["upsample_conv.bias"] = sd["model.classifier.4.bias"]
    return sd


def primary_keys(sd):
    """The keys that own storage (the `model.*` and `attention_module.*` names); the rest are the wrapper's aliases."""
    return [k for k in sd if k.startswith("model.") or k.startswith("attention_module.")]


def _bn(x, sd, p, training, relu=True):
    y = F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"],
                     training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if training:
        sd[f"{p}.num_batches_tracked"] += 1
    return torch.relu(y) if relu else y


def backbone_forward(sd, x, training, p="model.backbone"):
    """ResNet-50 stem + layer1..4 (torchvision resnet.py `_forward_impl` up to layer4; IntermediateLayerGetter 'out')."""
    x = F.conv2d(x, sd[f"{p}.conv1.weight"], None, stride=2, padding=3)              # :105-118 (1-channel stem)
    x = _bn(x, sd, f"{p}.bn1", training)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for name, _cin, _planes, stride, dil, ds in block_specs():
        q = f"{p}.{name}"
        idn = x
        out = _bn(F.conv2d(x, sd[f"{q}.conv1.weight"]), sd, f"{q}.bn1", training)
        out = _bn(F.conv2d(out, sd[f"{q}.conv2.weight"], None, stride=stride, padding=dil, dilation=dil), sd, f"{q}.bn2", training)
        out = _bn(F.conv2d(out, sd[f"{q}.conv3.weight"]), sd, f"{q}.bn3", training, relu=False)
        if ds:
            idn = _bn(F.conv2d(x, sd[f"{q}.downsample.0.weight"], None, stride=stride), sd, f"{q}.downsample.1", training, relu=False)
        x = torch.relu(out + idn)
    return x


def aspp_forward(sd, x, training, p="model.classifier.0", dropout_mask: Optional[torch.Tensor] = None):
    """torchvision deeplabv3.py ASPP: [1x1, 3x3 rate 12 / 24 / 36, image pooling] -> concat -> 1x1 -> BN -> ReLU -> Dropout(0.5).
    dropout_mask ([B,256,h,w] of 0/1, training only): the kept positions; None = no dropout (eval, or p = 0 tests)."""
    h, w = x.shape[-2:]
    outs = [_bn(F.conv2d(x, sd[f"{p}.convs.0.0.weight"]), sd, f"{p}.convs.0.1", training)]
    for i, r in enumerate(ASPP_RATES, start=1):
        outs.append(_bn(F.conv2d(x, sd[f"{p}.convs.{i}.0.weight"], None, padding=r, dilation=r), sd, f"{p}.convs.{i}.1", training))
    g = F.adaptive_avg_pool2d(x, 1)
    g = _bn(F.conv2d(g, sd[f"{p}.convs.4.1.weight"]), sd, f"{p}.convs.4.2", training)
    outs.append(F.interpolate(g, size=(h, w), mode="bilinear", align_corners=False))
    y = _bn(F.conv2d(torch.cat(outs, dim=1), sd[f"{p}.project.0.weight"]), sd, f"{p}.project.1", training)
    if training and dropout_mask is not None:
        y = y * dropout_mask.to(y.dtype) / (1.0 - DROPOUT_P)
    return y


def forward(sd, x, training: bool, dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DeepLabV3_SingleChannel_Attn.forward, DeepLabV3-ChannelAttention.py:140-162."""
    size = x.shape[-2:]                                                                   # :141
    f = backbone_forward(sd, x, training)                                                 # :144-145
    y = aspp_forward(sd, f, training, dropout_mask=dropout_mask)                          # :148
    y = _bn(F.conv2d(y, sd["model.classifier.1.weight"], None, padding=1), sd, "model.classifier.2", training)   # :151
    y = cam_layer(y, sd["attention_module.mlp.0.weight"], sd["attention_module.mlp.2.weight"])                  # :154
    y = F.conv2d(y, sd["model.classifier.4.weight"], sd["model.classifier.4.bias"])       # :157
    return F.interpolate(y, size=size, mode="bilinear", align_corners=False)              # :160 (resize, BILINEAR)
