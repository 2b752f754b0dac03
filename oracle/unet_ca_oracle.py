"""TEST INFRASTRUCTURE ONLY — CPU restatement of the U-Net-CA hot path.

A functional (state_dict-driven) restatement of the arithmetic of
`/root/reference/Unet-ChannalAttention.py`, written from the reference's
behaviour, used as the parity oracle for the HIP path and as the reported
`cpu_baseline` ("port") in bench.py. It is pinned against outputs of the
reference itself (imported in the build container with an inert torchvision
stub) by `tests/test_oracle_golden.py` using the fixtures under
`tests/golden/` produced by `oracle/gen_golden.py`.

Each function cites the reference lines it follows. All tensors are plain
torch CPU tensors; gradients come from torch autograd exactly as in the
reference (which trains with `loss.backward()`, `:345`).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default used at Unet-ChannalAttention.py:82,85
BN_MOMENTUM = 0.1
SE_REDUCTION = 16   # Unet-ChannalAttention.py:49
WIDTHS = (64, 128, 256, 512, 1024)  # Unet-ChannalAttention.py:105-109


# --------------------------------------------------------------------------------------
# state_dict contract: names, shapes and ORDER of the reference's UNet.
# --------------------------------------------------------------------------------------
def _double_conv_entries(prefix: str, cin: int, cout: int, use_se: bool):
    e = []
    for idx, ci in ((0, cin), (3, cout)):
        e.append((f"{prefix}.double_conv.{idx}.weight", (cout, ci, 3, 3)))
        e.append((f"{prefix}.double_conv.{idx}.bias", (cout,)))
        bn = idx + 1
        e.append((f"{prefix}.double_conv.{bn}.weight", (cout,)))
        e.append((f"{prefix}.double_conv.{bn}.bias", (cout,)))
        e.append((f"{prefix}.double_conv.{bn}.running_mean", (cout,)))
        e.append((f"{prefix}.double_conv.{bn}.running_var", (cout,)))
        e.append((f"{prefix}.double_conv.{bn}.num_batches_tracked", ()))
    if use_se:
        e.append((f"{prefix}.double_conv.6.fc.0.weight", (cout // SE_REDUCTION, cout)))
        e.append((f"{prefix}.double_conv.6.fc.2.weight", (cout, cout // SE_REDUCTION)))
    return e


def state_dict_template(in_channels: int = 1, num_classes: int = 2, use_se: bool = False):
    """Ordered (name -> zero tensor) matching `UNet.__init__` (Unet-ChannalAttention.py:100-125)."""
    w = WIDTHS
    entries = []
    entries += _double_conv_entries("inc", in_channels, w[0], use_se)
    for i in range(1, 5):
        entries += _double_conv_entries(f"down{i}.1", w[i - 1], w[i], use_se)
    # decoder: registration order in the reference is up1, conv1, up2, conv2, ...
    for i in range(1, 5):
        cin, cout = w[5 - i], w[4 - i]
        entries.append((f"up{i}.weight", (cin, cout, 2, 2)))
        entries.append((f"up{i}.bias", (cout,)))
        entries += _double_conv_entries(f"conv{i}", cin, cout, use_se)
    entries.append(("outc.weight", (num_classes, w[0], 1, 1)))
    entries.append(("outc.bias", (num_classes,)))
    sd = OrderedDict()
    for k, shp in entries:
        dt = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        sd[k] = torch.zeros(shp, dtype=dt)
    return sd


def is_param(name: str) -> bool:
    leaf = name.split(".")[-1]
    return leaf not in ("running_mean", "running_var", "num_batches_tracked")


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
# Test hook. The network has three kinds of discontinuous decisions: the ReLU after each BatchNorm, the ReLU
# inside the SE bottleneck and the arg-max of each 2x2 max-pool. When DECISION_HOOK is set to
# f(kind, name, tensor) -> tensor  (kind in "relu" | "se_relu" | "pool") it replaces that operation; the
# gradient parity tests use it to evaluate the oracle's backward under GIVEN decisions (y * mask, gather at
# given indices), which removes every discontinuity from the comparison (tests/test_parity_gpu.py).
DECISION_HOOK = None


def se_layer(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, name: str = "") -> torch.Tensor:
    """SELayer.forward, Unet-ChannalAttention.py:61-72 (two bias-free Linears, :54-59)."""
    b, c = x.shape[0], x.shape[1]
    squeeze = x.mean(dim=(2, 3))                        # AdaptiveAvgPool2d(1).view(b,c)  :65
    pre = squeeze @ w1.t()                              # Linear(C, C/16)                 :55
    hidden = DECISION_HOOK("se_relu", name, pre) if DECISION_HOOK is not None else torch.relu(pre)   # :56
    gate = torch.sigmoid(hidden @ w2.t())               # Linear(C/16, C) + Sigmoid       :57-58
    return x * gate.view(b, c, 1, 1)                    # :72


def cam_layer(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """ChannelAttentionModule.forward, DeepLabV3-ChannelAttention.py:66-79: a shared bias-free MLP of two 1x1
    convolutions (:59-63; weights (C/r, C, 1, 1) and (C, C/r, 1, 1)) on the spatial mean and on the spatial
    maximum, summed, sigmoid, times x."""
    b, c = x.shape[0], x.shape[1]
    w1m, w2m = w1.reshape(w1.shape[0], c), w2.reshape(c, w2.shape[1])
    avg = x.mean(dim=(2, 3))                                  # AdaptiveAvgPool2d(1)   :68
    mx = x.amax(dim=(2, 3))                                   # AdaptiveMaxPool2d(1)   :69
    out = torch.relu(avg @ w1m.t()) @ w2m.t() + torch.relu(mx @ w1m.t()) @ w2m.t()    # :72
    return x * torch.sigmoid(out).view(b, c, 1, 1)            # :75-78


def _bn_relu(x, sd, prefix, training, eps, momentum):
    """BatchNorm2d (+ReLU), Unet-ChannalAttention.py:82-83,85-86. Updates running stats in `sd`
    (momentum 0.1, unbiased variance) when training, like nn.BatchNorm2d."""
    y = F.batch_norm(
        x, sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"],
        sd[f"{prefix}.weight"], sd[f"{prefix}.bias"],
        training=training, momentum=momentum, eps=eps)
    if training:
        sd[f"{prefix}.num_batches_tracked"] += 1
    if DECISION_HOOK is not None:
        return DECISION_HOOK("relu", prefix, y)
    return torch.relu(y)


def double_conv(x, sd, prefix: str, use_se: bool, training: bool,
                eps: float = BN_EPS, momentum: float = BN_MOMENTUM):
    """DoubleConv.forward, Unet-ChannalAttention.py:75-97."""
    p = f"{prefix}.double_conv"
    x = F.conv2d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], padding=1)
    x = _bn_relu(x, sd, f"{p}.1", training, eps, momentum)
    x = F.conv2d(x, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"], padding=1)
    x = _bn_relu(x, sd, f"{p}.4", training, eps, momentum)
    if use_se:
        x = se_layer(x, sd[f"{p}.6.fc.0.weight"], sd[f"{p}.6.fc.2.weight"], f"{p}.6")
    return x


def unet_forward(sd, x: torch.Tensor, use_se: bool = True, training: bool = True) -> torch.Tensor:
    """UNet.forward, Unet-ChannalAttention.py:127-163. For H, W multiples of 16 the bilinear-resize fallback
    (:138-139,144-145,150-151,156-157) is never taken and everything here is pinned by the golden vectors. For other sizes
    the fallback `F_T.resize(x, size, interpolation=BILINEAR)` runs on a tensor: torchvision implements that as
    torch.nn.functional.interpolate(x, size, mode="bilinear", align_corners=False[, antialias]) — antialiasing only acts
    when DOWN-scaling, and here the map grows by one pixel — restated below; torchvision is absent from the build container
    (SURVEY 8c), so this branch is parity-UNPINNED."""
    skips = []
    h = double_conv(x, sd, "inc", use_se, training)
    for i in range(1, 5):
        skips.append(h)
        h = DECISION_HOOK("pool", f"down{i}.0", h) if DECISION_HOOK is not None else F.max_pool2d(h, 2)   # :106-109
        h = double_conv(h, sd, f"down{i}.1", use_se, training)
    for i in range(1, 5):
        h = F.conv_transpose2d(h, sd[f"up{i}.weight"], sd[f"up{i}.bias"], stride=2)   # :112..
        if h.shape[2:] != skips[4 - i].shape[2:]:                      # :138-139 (resize fallback)
            h = F.interpolate(h, size=tuple(skips[4 - i].shape[2:]), mode="bilinear", align_corners=False)
        h = torch.cat([skips[4 - i], h], dim=1)                        # skip first, :140
        h = double_conv(h, sd, f"conv{i}", use_se, training)
    return F.conv2d(h, sd["outc.weight"], sd["outc.bias"])            # :125,162


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = 255) -> torch.Tensor:
    """nn.CrossEntropyLoss(ignore_index=255), Unet-ChannalAttention.py:465,344: mean over
    non-ignored pixels of -log softmax(logits)[target]."""
    logp = torch.log_softmax(logits, dim=1)
    valid = target != ignore_index
    safe = torch.where(valid, target, torch.zeros_like(target))
    picked = logp.gather(1, safe.unsqueeze(1)).squeeze(1)
    return -(picked * valid).sum() / valid.sum()


def soft_dice_loss(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = 255,
                   smooth: float = 1.0) -> torch.Tensor:
    """Standard soft-Dice on softmax probabilities, mean over classes. The reference has NO
    Dice loss; this is the definition the build-side DiceLoss is checked against."""
    c = logits.shape[1]
    prob = torch.softmax(logits, dim=1)
    valid = (target != ignore_index)
    safe = torch.where(valid, target, torch.zeros_like(target))
    onehot = F.one_hot(safe, c).permute(0, 3, 1, 2).to(prob.dtype) * valid.unsqueeze(1)
    prob = prob * valid.unsqueeze(1)
    inter = (prob * onehot).sum(dim=(0, 2, 3))
    denom = prob.sum(dim=(0, 2, 3)) + onehot.sum(dim=(0, 2, 3))
    return 1.0 - ((2 * inter + smooth) / (denom + smooth)).mean()


# --------------------------------------------------------------------------------------
# optimizer: optim.Adam(lr=1e-4) defaults, Unet-ChannalAttention.py:466
# --------------------------------------------------------------------------------------
def adam_update(params: List[torch.Tensor], grads: List[torch.Tensor], state: Dict[int, dict],
                lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor semantics (no weight decay / amsgrad): eps is added to
    sqrt(v)/sqrt(bias_correction2), step size lr/bias_correction1."""
    b1, b2 = betas
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            st = state.setdefault(i, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            st["m"].lerp_(g, 1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            denom = (st["v"].sqrt() / (bc2 ** 0.5)).add_(eps)
            p.addcdiv_(st["m"], denom, value=-lr / bc1)


def train_step(sd, opt_state, x, target, use_se=True, lr=1e-4, dice_weight: float = 0.0) -> Tuple[float, torch.Tensor]:
    """One iteration of the reference's hot loop, Unet-ChannalAttention.py:342-346:
    zero_grad -> forward -> CE -> backward -> Adam.step. Mutates `sd` / `opt_state`."""
    names = [k for k in sd if is_param(k)]
    leaves = []
    work = OrderedDict(sd)
    for k in names:
        leaf = sd[k].detach().requires_grad_(True)
        work[k] = leaf
        leaves.append(leaf)
    logits = unet_forward(work, x, use_se=use_se, training=True)
    loss = cross_entropy(logits, target)
    if dice_weight:
        loss = loss + dice_weight * soft_dice_loss(logits, target)
    grads = torch.autograd.grad(loss, leaves)
    for k in sd:                                      # BN buffers were updated in `work`
        if not is_param(k):
            sd[k] = work[k]
    plist = [sd[k] for k in names]
    adam_update(plist, list(grads), opt_state, lr=lr)
    return float(loss.detach()), logits.detach()


# --------------------------------------------------------------------------------------
# metrics: compute_metrics, Unet-ChannalAttention.py:215-269 (quirks kept)
# --------------------------------------------------------------------------------------
def confusion_counts(logits: torch.Tensor, masks: torch.Tensor, num_classes: int):
    """TP/FP/FN per class over non-255 pixels (:220-240). argmax ties -> lower index (:220)."""
    preds = torch.max(logits, 1)[1]
    valid = masks != 255
    p = preds[valid].cpu().numpy()
    m = masks[valid].cpu().numpy()
    tp = np.zeros(num_classes)
    fp = np.zeros(num_classes)
    fn = np.zeros(num_classes)
    for c in range(num_classes):
        tp[c] = ((m == c) & (p == c)).sum()
        fp[c] = ((m != c) & (p == c)).sum()
        fn[c] = ((m == c) & (p != c)).sum()
    return tp, fp, fn


def metrics_from_counts(tp, fp, fn) -> Dict[str, float]:
    """acc = sum TP / (sum TP + sum FP + sum FN) (:243-245, NOT pixel accuracy); miou over
    classes with union>0 (:248-251); mpa/mf1 over classes present in GT (:254-262)."""
    tp, fp, fn = (np.asarray(a, dtype=float) for a in (tp, fp, fn))
    tot = tp.sum() + fp.sum() + fn.sum()
    acc = tp.sum() / tot if tot > 0 else 0.0
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


def compute_metrics(logits, masks, num_classes: int = 2) -> Dict[str, float]:
    return metrics_from_counts(*confusion_counts(logits, masks, num_classes))
