"""TEST INFRASTRUCTURE ONLY — G7: golden vectors for ChannelAttentionModule (config 5), generated from the
reference's own class (DeepLabV3-ChannelAttention.py:49-79, imported with inert torchvision stubs).
Re-run with:  python -m oracle.gen_golden_cam"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import closed_form as cf
from .gen_golden import OUT, block_fixture
from .ref_loader import load_reference_deeplab_ca


def main() -> None:
    ref = load_reference_deeplab_ca()
    g7 = {}
    block_fixture(ref.ChannelAttentionModule(256, 16), cf.make_input((2, 256, 8, 8), 0.4), True, g7, "cam256")
    x = cf.make_input((3, 64, 12, 20), 0.9)
    x[0, 5, 2, 3] = x[0, 5, 7, 11] = 2.5          # an exact tie for the maximum: the first in scan order gets the gradient
    x[2, :, 4, 4] = 3.0                           # one pixel holds the maximum of every channel
    g7["cam64_ties/x"] = x.numpy().copy()
    block_fixture(ref.ChannelAttentionModule(64, 16), x, True, g7, "cam64_ties")
    np.savez_compressed(os.path.join(OUT, "g7_cam.npz"), **g7)
    print("wrote g7_cam.npz:", len(g7), "arrays")


if __name__ == "__main__":
    main()
