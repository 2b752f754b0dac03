"""Training / validation loop with the reference's entry points and bookkeeping
(`compute_metrics`, `validate_model`, `train_model`: Unet-ChannalAttention.py:215-269, 273-317, 321-399),
kept on the device: the reference synchronises every step (`loss.item()` :348 and a 16 MB device-to-host
copy of predictions and masks inside compute_metrics :229-230); here the per-batch loss and the per-batch
TP/FP/FN counts (`insar_confusion`) stay in HBM and are read back once per epoch.

Semantics kept on purpose: metrics are computed PER BATCH and averaged with sample weights
(not from a dataset-level confusion matrix); `acc` is sum TP / (sum TP + sum FP + sum FN); train metrics
divide by len(dataset); the history list has the reference's keys; the best-val-mIoU state_dict is saved.
"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr


def metrics_from_counts(tp, fp, fn) -> Dict[str, float]:
    """The arithmetic of compute_metrics (:243-262) on per-class TP/FP/FN counts."""
    tp, fp, fn = (np.asarray(a, dtype=float) for a in (tp, fp, fn))
    total = tp.sum() + fp.sum() + fn.sum()
    acc = tp.sum() / total if total > 0 else 0.0
    union = tp + fp + fn
    iou = np.divide(tp, union, out=np.zeros_like(tp), where=union != 0)
    miou = float(np.mean(iou[union > 0])) if np.any(union > 0) else 0.0
    gt = tp + fn
    recall = np.divide(tp, gt, out=np.zeros_like(tp), where=gt != 0)
    mpa = float(np.mean(recall[gt > 0])) if np.any(gt > 0) else 0.0
    pp = tp + fp
    precision = np.divide(tp, pp, out=np.zeros_like(tp), where=pp != 0)
    pr = precision + recall
    f1 = np.divide(2 * precision * recall, pr, out=np.zeros_like(tp), where=pr != 0)
    mf1 = float(np.mean(f1[gt > 0])) if np.any(gt > 0) else 0.0
    return {"acc": float(acc), "miou": miou, "mpa": mpa, "mf1": mf1}


def confusion_counts(outputs: torch.Tensor, masks: torch.Tensor, num_classes: int,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int64 [3, num_classes] (TP, FP, FN) on the device; argmax ties go to the lower class, 255 is ignored."""
    if not outputs.is_cuda:
        raise _lib.InsarError("confusion_counts: logits must be a ROCm tensor (no CPU fallback)")
    lg = outputs.detach()
    if lg.dtype != torch.float32 or not lg.is_contiguous():
        lg = lg.float().contiguous()
    mk = masks if (masks.dtype == torch.int64 and masks.is_contiguous()) else masks.long().contiguous()
    if out is None:
        out = torch.empty(3, num_classes, dtype=torch.int64, device=lg.device)
    hw = 1
    for d in lg.shape[2:]:
        hw *= d
    call("insar_confusion", ptr(lg), ptr(mk), lg.shape[0], num_classes, hw, 255, ptr(out), _lib.stream_ptr())
    return out


def compute_metrics(outputs: torch.Tensor, masks: torch.Tensor, num_classes: int) -> Dict[str, float]:
    """Drop-in for the reference's compute_metrics (:215-269); synchronises (reads 3*num_classes ints)."""
    c = confusion_counts(outputs, masks, num_classes).cpu().numpy()
    return metrics_from_counts(c[0], c[1], c[2])


def _epoch_metrics(counts: List[torch.Tensor], sizes: List[int]) -> Dict[str, float]:
    tot = {"acc": 0.0, "miou": 0.0, "mpa": 0.0, "mf1": 0.0}
    if not counts:
        return tot
    allc = torch.stack(counts).cpu().numpy()            # one device-to-host copy per epoch
    for c, n in zip(allc, sizes):
        m = metrics_from_counts(c[0], c[1], c[2])
        for k in tot:
            tot[k] += m[k] * n
    return tot


@torch.no_grad()
def validate_model(model, dataloader, criterion, device, num_classes: int = 2, verbose: bool = True) -> Dict[str, float]:
    """validate_model (:273-317): eval-mode pass, sample-weighted mean of per-batch loss and metrics."""
    was_training = model.training
    model.eval()
    losses, counts, sizes = [], [], []
    for images, masks in dataloader:
        images = images.to(device, non_blocking=True)
        masks = masks.to(device, non_blocking=True)
        outputs = model(images)
        losses.append(criterion(outputs, masks).detach().float() * images.size(0))
        counts.append(confusion_counts(outputs, masks, num_classes))
        sizes.append(images.size(0))
    n = sum(sizes)
    if n > 0:
        tot = _epoch_metrics(counts, sizes)
        res = {"val_loss": float(torch.stack(losses).sum()) / n, "val_acc": tot["acc"] / n, "val_miou": tot["miou"] / n,
               "val_mpa": tot["mpa"] / n, "val_mf1": tot["mf1"] / n}
    else:
        res = {"val_loss": 0.0, "val_acc": 0.0, "val_miou": 0.0, "val_mpa": 0.0, "val_mf1": 0.0}
    if verbose:
        print(f"val loss {res['val_loss']:.4f} acc {res['val_acc']:.4f} mIoU {res['val_miou']:.4f} "
              f"mPA {res['val_mpa']:.4f} mF1 {res['val_mf1']:.4f}")
    model.train(was_training)
    return res


def save_history(history: List[Dict[str, Any]], path: str) -> None:
    """The reference's metrics file (:472-487): a JSON list with one dict per epoch (`epoch`, `train_*`,
    `val_*` keys), tensors converted to Python floats, indent 4."""
    import json
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    rows = [{k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in rec.items()} for rec in history]
    with open(path, "w") as f:
        json.dump(rows, f, indent=4)


def train_model(model, train_dataloader, val_dataloader, criterion, optimizer, device, num_epochs: int = 25,
                num_classes: int = 2, model_save_path: Optional[str] = None, verbose: bool = True) -> List[Dict[str, Any]]:
    """train_model (:321-399): returns the per-epoch history (train_* / val_* keys of the reference);
    saves `model.state_dict()` whenever the validation mIoU improves (:382-387) if a path is given."""
    model.to(device)
    start = time.time()
    best = -1.0
    history: List[Dict[str, Any]] = []
    for epoch in range(num_epochs):
        model.train()
        losses, counts, sizes = [], [], []
        for images, masks in train_dataloader:
            images = images.to(device, non_blocking=True)
            masks = masks.to(device, non_blocking=True)
            optimizer.zero_grad()
            outputs = model(images)
            loss = criterion(outputs, masks)
            loss.backward()
            optimizer.step()
            losses.append(loss.detach().float() * images.size(0))
            counts.append(confusion_counts(outputs, masks, num_classes))
            sizes.append(images.size(0))
        n_train = len(train_dataloader.dataset) if hasattr(train_dataloader, "dataset") else sum(sizes)
        tot = _epoch_metrics(counts, sizes)
        rec: Dict[str, Any] = {"epoch": epoch + 1, "train_loss": float(torch.stack(losses).sum()) / n_train,
                               "train_acc": tot["acc"] / n_train, "train_miou": tot["miou"] / n_train,
                               "train_mpa": tot["mpa"] / n_train, "train_mf1": tot["mf1"] / n_train}
        if verbose:
            print(f"epoch {epoch + 1}/{num_epochs} train loss {rec['train_loss']:.4f} acc {rec['train_acc']:.4f} "
                  f"mIoU {rec['train_miou']:.4f}")
        if val_dataloader:
            val = validate_model(model, val_dataloader, criterion, device, num_classes, verbose)
            rec.update(val)
            if val["val_miou"] > best:
                best = val["val_miou"]
                if model_save_path:
                    os.makedirs(os.path.dirname(model_save_path) or ".", exist_ok=True)
                    torch.save(model.state_dict(), model_save_path)
        history.append(rec)
    if verbose:
        print(f"training finished in {(time.time() - start) / 60:.2f} min")
    return history
