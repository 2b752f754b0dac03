"""TEST INFRASTRUCTURE ONLY — import the *reference* script in the build container.

`/root/reference` exists only in the build container (never on the GPU box),
so this module is used solely by `oracle/gen_golden.py` (fixture generation)
and by the in-container oracle-vs-reference test, which skips when the
reference tree is absent. torchvision is not installed here; the reference
imports it at module top (Unet-ChannalAttention.py:5,7) but the hot path never
calls it for H, W multiples of 16, so inert stub modules are registered first.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "Unet-ChannalAttention.py"))


def _install_torchvision_stub() -> None:
    if "torchvision" in sys.modules:
        return

    def _unavailable(*_a, **_k):
        raise NotImplementedError("torchvision is not installed; inert stub")

    class InterpolationMode:
        BILINEAR = "bilinear"
        NEAREST = "nearest"

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    for name in ("Compose", "Resize", "ToTensor", "Normalize"):
        setattr(tr, name, _unavailable)
    tr.InterpolationMode = InterpolationMode
    fn.resize = _unavailable
    tv.transforms = tr
    tr.functional = fn
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr
    sys.modules["torchvision.transforms.functional"] = fn


def load_reference_unet_ca():
    """Return the reference module object for Unet-ChannalAttention.py."""
    _install_torchvision_stub()
    path = os.path.join(REFERENCE_ROOT, "Unet-ChannalAttention.py")
    spec = importlib.util.spec_from_file_location("ref_unet_ca", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_deeplab_ca():
    """Return the reference module object for DeepLabV3-ChannelAttention.py (config 5). Only its
    ChannelAttentionModule (:49-79) is usable here: the DeepLab constructor needs torchvision and fetches
    ImageNet weights and is never called. The import prints the device (:41-45)."""
    _install_torchvision_stub()
    if "torchvision.models" not in sys.modules:
        def _unavailable(*_a, **_k):
            raise NotImplementedError("torchvision is not installed; inert stub")
        models = types.ModuleType("torchvision.models")
        seg = types.ModuleType("torchvision.models.segmentation")
        seg.deeplabv3_resnet50 = _unavailable
        seg.deeplabv3_resnet101 = _unavailable
        models.segmentation = seg
        sys.modules["torchvision"].models = models
        sys.modules["torchvision.models"] = models
        sys.modules["torchvision.models.segmentation"] = seg
    path = os.path.join(REFERENCE_ROOT, "DeepLabV3-ChannelAttention.py")
    spec = importlib.util.spec_from_file_location("ref_deeplab_ca", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
