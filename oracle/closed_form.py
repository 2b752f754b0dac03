"""TEST INFRASTRUCTURE ONLY — deterministic closed-form fillers.

Weights and inputs used by the golden fixtures must not depend on torch's
RNG (its stream differs across versions/devices), so every tensor is a
closed-form function of (tensor ordinal, flat element index), evaluated in
float64 with numpy and rounded once to float32. Both the fixture generator
(`oracle/gen_golden.py`, which feeds the *reference* classes) and the parity
tests (which feed the product) call these same functions.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


def _wave(n: int, ordinal: int, salt: float) -> np.ndarray:
    """Pseudo-random looking but closed-form values in [-1, 1]."""
    i = np.arange(n, dtype=np.float64)
    # two incommensurate frequencies; arguments kept small enough for exact-ish fp64 sin
    a = np.sin(i * 0.754877666 + ordinal * 1.3247179 + salt)
    b = np.sin(i * 0.569840291 * 1.618033989 + ordinal * 0.4142135 + 2.0 * salt)
    return 0.6 * a + 0.4 * b


def fill_tensor(name: str, shape, ordinal: int) -> torch.Tensor:
    """Closed-form value for one state_dict entry (by name suffix + shape)."""
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    w = _wave(n, ordinal, 0.25)
    if leaf == "running_mean":
        v = 0.05 * w
    elif leaf == "running_var":
        v = 1.0 + 0.2 * w
    elif leaf == "weight" and len(shape) == 1:      # BN gamma
        v = 1.0 + 0.15 * w
    elif leaf == "bias":                            # conv / BN bias
        v = 0.1 * w
    elif leaf == "weight":
        if len(shape) == 4:
            # Conv2d (Co,Ci,kh,kw): fan_in = Ci*kh*kw ; ConvTranspose2d (Ci,Co,kh,kw): use Ci*kh*kw too
            fan_in = shape[1] * shape[2] * shape[3] if ".up" not in "." + name else shape[0]
        else:
            fan_in = shape[1]
        v = w * math.sqrt(3.0 / max(fan_in, 1)) * 1.4
    else:
        v = 0.1 * w
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def fill_state_dict(template: "OrderedDict[str, torch.Tensor]") -> "OrderedDict[str, torch.Tensor]":
    """Return a new state_dict with the same keys/shapes filled in closed form."""
    out = OrderedDict()
    for ordinal, (k, t) in enumerate(template.items()):
        out[k] = fill_tensor(k, t.shape, ordinal)
    return out


def make_input(shape, salt: float = 0.0) -> torch.Tensor:
    """Closed-form [B,C,H,W] input in roughly [-1,1] (the reference's post-Normalize range)."""
    b, c, h, w = shape
    n = np.arange(b, dtype=np.float64)[:, None, None, None]
    ch = np.arange(c, dtype=np.float64)[None, :, None, None]
    y = np.arange(h, dtype=np.float64)[None, None, :, None]
    x = np.arange(w, dtype=np.float64)[None, None, None, :]
    phase = 0.173 * y * (1.0 + 0.31 * ch) + 0.097 * x * (1.0 + 0.17 * n) + 1.7 * n + 0.9 * ch + salt
    v = 0.7 * np.sin(phase) + 0.3 * np.cos(0.011 * x * y + 0.5 * ch + 2.1 * n + salt)
    return torch.from_numpy(v.astype(np.float32))


def make_grad(shape, salt: float = 0.5) -> torch.Tensor:
    """Closed-form upstream gradient of the given shape."""
    n = int(np.prod(shape))
    return torch.from_numpy((_wave(n, 7, salt) * 0.5).reshape(tuple(shape)).astype(np.float32))


def make_target(shape, ignore_every: int = 0) -> torch.Tensor:
    """Closed-form int64 {0,1} segmentation target [B,H,W]; optional 255 entries."""
    b, h, w = shape
    n = np.arange(b)[:, None, None]
    y = np.arange(h)[None, :, None]
    x = np.arange(w)[None, None, :]
    t = (((x * 3 + y * 5 + n * 7) % 11) < 3).astype(np.int64)
    if ignore_every:
        t = np.where(((x + 2 * y + n) % ignore_every) == 0, 255, t)
    return torch.from_numpy(t)


def sample_indices(numel: int, k: int = 16) -> np.ndarray:
    """k deterministic flat indices spread over [0, numel)."""
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * 2654435761 % numel).astype(np.int64)


# ---- generic-position fixtures (numpy PCG64) ---------------------------------------------------------
# The closed-form (sine) fillers above give smooth, highly correlated activations: many ReLU inputs sit
# within rounding of 0 and gradients become ill-conditioned (torch's own fp32 and fp64 gradients differ by
# 3-15 % on them). Gradient parity is therefore pinned on random, generic-position weights and inputs
# drawn from numpy's PCG64 stream (stable across machines; no torch RNG), where the same comparison
# agrees to ~3e-4.
def fill_state_dict_random(template: "OrderedDict[str, torch.Tensor]", seed: int = 7) -> "OrderedDict[str, torch.Tensor]":
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for k, t in template.items():
        leaf = k.split(".")[-1]
        shape = tuple(t.shape)
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros((), dtype=torch.int64)
        elif leaf == "running_var":
            out[k] = torch.from_numpy((1.0 + 0.2 * rng.random(shape)).astype(np.float32))
        elif leaf == "running_mean":
            out[k] = torch.from_numpy((0.05 * rng.standard_normal(shape)).astype(np.float32))
        elif leaf == "weight" and len(shape) == 1:
            out[k] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif leaf == "bias":
            out[k] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            out[k] = torch.from_numpy((rng.standard_normal(shape) * math.sqrt(2.0 / max(fan_in, 1))).astype(np.float32))
    return out


def make_input_random(shape, seed: int = 11) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


def make_target_random(shape, seed: int = 13, ignore_frac: float = 0.0) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64(seed))
    t = rng.integers(0, 2, size=tuple(shape)).astype(np.int64)
    if ignore_frac > 0:
        t = np.where(rng.random(tuple(shape)) < ignore_frac, 255, t)
    return torch.from_numpy(t)
