"""Synthetic InSAR tiles: wrapped interferometric phase with linear deformation
features, delivered as (cos phi, sin phi) channels in [-1, 1] — the range the reference's
Normalize(0.5, 0.5) produces (Unet-ChannalAttention.py:431) — plus a {0,1} int64 label that marks
pixels within 2 px of a feature line. Seeded with numpy PCG64 only (no torch RNG), so the same
tile index gives the same tile on every machine."""
from __future__ import annotations

import numpy as np
import torch

TRAIN_SEED0 = 1000
HELDOUT_SEED0 = 900000


def make_tile(seed: int, size: int = 256, channels: int = 2):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = w = size
    v, u = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    a, b = rng.uniform(-8 * np.pi, 8 * np.pi, size=2) / size
    phi = a * u + b * v
    label = np.zeros((h, w), dtype=np.int64)
    for _ in range(int(rng.integers(1, 4))):
        x0, y0, x1, y1 = rng.uniform(0, size, size=4)
        step = rng.uniform(np.pi / 2, 2 * np.pi)
        dx, dy = x1 - x0, y1 - y0
        ln = max(np.hypot(dx, dy), 1e-6)
        side = ((u - x0) * dy - (v - y0) * dx) / ln          # signed distance to the infinite line
        t = ((u - x0) * dx + (v - y0) * dy) / (ln * ln)      # position along the segment
        phi = phi + step * (side > 0) * ((t >= 0) & (t <= 1))
        tc = np.clip(t, 0.0, 1.0)
        dist = np.hypot(u - (x0 + tc * dx), v - (y0 + tc * dy))
        label[dist <= 2.0] = 1
    phi = phi + 0.3 * rng.standard_normal((h, w))
    phi = np.angle(np.exp(1j * phi))
    if channels == 2:
        img = np.stack([np.cos(phi), np.sin(phi)], 0)
    else:
        img = (phi / np.pi)[None]
    return img.astype(np.float32), label


def make_batch(first_index: int, batch: int, size: int = 256, heldout: bool = False, channels: int = 2):
    seed0 = HELDOUT_SEED0 if heldout else TRAIN_SEED0
    tiles = [make_tile(seed0 + first_index + i, size, channels) for i in range(batch)]
    x = torch.from_numpy(np.stack([t[0] for t in tiles], 0))
    y = torch.from_numpy(np.stack([t[1] for t in tiles], 0))
    return x, y


class SyntheticTiles(torch.utils.data.Dataset):
    """Dataset with the reference's (img [C,S,S] float32 in [-1,1], mask [S,S] int64) contract
    (VOCSegDataset.__getitem__, Unet-ChannalAttention.py:191-212) over the synthetic generator."""

    def __init__(self, count: int, size: int = 256, heldout: bool = False, channels: int = 2, offset: int = 0):
        self.count, self.size, self.heldout, self.channels, self.offset = count, size, heldout, channels, offset

    def __len__(self):
        return self.count

    def __getitem__(self, idx: int):
        seed0 = HELDOUT_SEED0 if self.heldout else TRAIN_SEED0
        img, lab = make_tile(seed0 + self.offset + idx, self.size, self.channels)
        return torch.from_numpy(img), torch.from_numpy(lab)
