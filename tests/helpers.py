"""Shared comparison helpers for the parity tests."""
from __future__ import annotations

import numpy as np
import torch

from oracle import closed_form as cf


def to_np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().double().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def max_rel(a, b) -> float:
    """max |a-b| / max|b| (the 'max-rel' figure of SURVEY §8d)."""
    a, b = to_np(a), to_np(b)
    den = np.abs(b).max()
    if den == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / den)


def check_summary(store, prefix: str, t, rtol: float, what: str = "") -> None:
    """Compare tensor `t` against a fixture written by oracle.gen_golden.summarize()."""
    a = to_np(t).reshape(-1)
    absmax = float(store[f"{prefix}/absmax"])
    scale = max(absmax, 1e-30)
    idx = cf.sample_indices(a.size, 64)
    exp = store[f"{prefix}/samples"].astype(np.float64)
    err = np.abs(a[idx] - exp).max() / scale
    assert err <= rtol, f"{what or prefix}: sampled max-rel {err:.3e} > {rtol:.1e}"
    nrm = float(store[f"{prefix}/norm"])
    got = float(np.sqrt((a * a).sum()))
    assert abs(got - nrm) <= rtol * max(nrm, 1e-30) * 4 + 1e-12, \
        f"{what or prefix}: norm {got:.6e} vs {nrm:.6e}"
    key = f"{prefix}/full"
    if key in store.files:
        full = store[key].astype(np.float64).reshape(-1)
        err = np.abs(a - full).max() / scale
        assert err <= rtol, f"{what or prefix}: full max-rel {err:.3e} > {rtol:.1e}"


def check_grad_summary(store, prefix: str, t, rtol: float, outlier_frac: float = 2e-3, what: str = "") -> None:
    """Gradient comparison that tolerates isolated ReLU-mask flips: a pre-activation that sits within
    rounding of 0 may get a different mask in two fp32 implementations, which changes ONE element of a
    BN/bias gradient (or a thin slice of a weight gradient) by O(1) while everything else agrees.
    Rules: the norm agrees to 4*rtol; at most one of the 64 sampled elements may miss rtol; for tensors
    stored whole, at most `outlier_frac` of the elements (and at least one) may miss rtol."""
    a = to_np(t).reshape(-1)
    scale = max(float(store[f"{prefix}/absmax"]), 1e-30)
    idx = cf.sample_indices(a.size, 64)
    exp = store[f"{prefix}/samples"].astype(np.float64)
    bad = int((np.abs(a[idx] - exp) / scale > rtol).sum())
    assert bad <= 1, f"{what or prefix}: {bad} of {len(idx)} sampled elements off by more than {rtol:.1e}"
    nrm = float(store[f"{prefix}/norm"])
    got = float(np.sqrt((a * a).sum()))
    assert abs(got - nrm) <= 4 * rtol * max(nrm, 1e-30) + 1e-12, f"{what or prefix}: norm {got:.6e} vs {nrm:.6e}"
    key = f"{prefix}/full"
    if key in store.files:
        full = store[key].astype(np.float64).reshape(-1)
        nbad = int((np.abs(a - full) / scale > rtol).sum())
        assert nbad <= max(1, int(outlier_frac * a.size)), \
            f"{what or prefix}: {nbad} of {a.size} elements off by more than {rtol:.1e}"
