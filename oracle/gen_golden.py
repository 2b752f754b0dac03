"""TEST INFRASTRUCTURE ONLY — generate the golden fixtures under tests/golden/.

Runs the *reference* implementation (imported from /root/reference with an
inert torchvision stub, see ref_loader.py) on closed-form weights/inputs
(closed_form.py) and stores inputs' recipes + expected outputs as small .npz
fixtures. The reference itself never leaves the build container; only these
vectors do. Re-run with:  python -m oracle.gen_golden
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

from . import closed_form as cf
from .ref_loader import load_reference_unet_ca

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FULL_LIMIT = 16384      # tensors up to this many elements are stored whole
NSAMPLE = 64


def summarize(prefix: str, t: torch.Tensor, store: dict) -> None:
    a = t.detach().double().numpy().reshape(-1)
    store[f"{prefix}/norm"] = np.array(np.sqrt((a * a).sum()))
    store[f"{prefix}/sum"] = np.array(a.sum())
    store[f"{prefix}/absmax"] = np.array(np.abs(a).max() if a.size else 0.0)
    idx = cf.sample_indices(a.size, NSAMPLE)
    store[f"{prefix}/samples"] = a[idx].astype(np.float32)
    if a.size <= FULL_LIMIT:
        store[f"{prefix}/full"] = t.detach().float().numpy().copy()


def load_closed_form(module: nn.Module) -> None:
    module.load_state_dict(cf.fill_state_dict(module.state_dict()))


def block_fixture(module: nn.Module, x: torch.Tensor, training: bool, store: dict, tag: str,
                  steps: int = 1, need_dx: bool = True) -> None:
    load_closed_form(module)
    module.train(training)
    for s in range(steps):
        xin = x.clone().requires_grad_(need_dx)
        for p in module.parameters():
            p.grad = None
        out = module(xin)
        g = cf.make_grad(out.shape)
        out.backward(g)
        sfx = f"{tag}/step{s}"
        summarize(f"{sfx}/out", out, store)
        if need_dx:
            summarize(f"{sfx}/dx", xin.grad, store)
        for name, p in module.named_parameters():
            summarize(f"{sfx}/grad/{name}", p.grad, store)
        for name, b in module.named_buffers():
            summarize(f"{sfx}/buf/{name}", b.double() if b.dtype != torch.float32 else b, store)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(max(1, (os.cpu_count() or 2)))
    ref = load_reference_unet_ca()

    # ---- G1: blocks -------------------------------------------------------------------
    g1 = {}
    block_fixture(ref.SELayer(64), cf.make_input((2, 64, 8, 8)), True, g1, "se64")
    block_fixture(ref.SELayer(128), cf.make_input((3, 128, 4, 4), 0.3), True, g1, "se128")
    block_fixture(ref.DoubleConv(2, 64, use_se=True), cf.make_input((2, 2, 16, 16)), True, g1,
                  "dc_2_64_se_train", steps=2, need_dx=False)
    block_fixture(ref.DoubleConv(64, 128, use_se=True), cf.make_input((2, 64, 16, 16)), True, g1,
                  "dc_64_128_se_train", steps=2)
    block_fixture(ref.DoubleConv(64, 128, use_se=True), cf.make_input((2, 64, 16, 16)), False, g1,
                  "dc_64_128_se_eval")
    block_fixture(ref.DoubleConv(128, 64, use_se=False), cf.make_input((2, 128, 16, 16)), True, g1,
                  "dc_128_64_plain_train")
    block_fixture(ref.DoubleConv(128, 64, use_se=True), cf.make_input((1, 128, 8, 24), 0.7), True, g1,
                  "dc_128_64_se_ragged")          # M = 192 pixels: exercises a partial GEMM tile
    np.savez_compressed(os.path.join(OUT, "g1_blocks.npz"), **g1)

    # ---- G2: resample ops ---------------------------------------------------------------
    g2 = {}
    xp = cf.make_input((2, 64, 8, 8), 0.2)
    xp[0, :, 0:2, 0:2] = 0.25            # a 2x2 window of exact ties: grad goes to the first max
    xp[1, 3, 4:6, 2:4] = -0.5
    xin = xp.clone().requires_grad_(True)
    pooled = nn.MaxPool2d(2)(xin)
    pooled.backward(cf.make_grad(pooled.shape))
    g2["pool/x"] = xp.numpy()
    summarize("pool/out", pooled, g2)
    summarize("pool/dx", xin.grad, g2)
    block_fixture(nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2), cf.make_input((2, 128, 8, 8)),
                  True, g2, "convT_128_64")
    block_fixture(nn.Conv2d(64, 2, kernel_size=1), cf.make_input((2, 64, 16, 16)), True, g2, "outc_64_2")
    np.savez_compressed(os.path.join(OUT, "g2_resample.npz"), **g2)

    # ---- G3: whole network ---------------------------------------------------------------
    g3 = {}
    crit = nn.CrossEntropyLoss(ignore_index=255)
    for tag, shape, training in (("b2_64_train", (2, 2, 64, 64), True),
                                 ("b1_256_train", (1, 2, 256, 256), True),
                                 ("b1_256_eval", (1, 2, 256, 256), False),
                                 ("b3_48x80_train", (3, 2, 48, 80), True)):
        net = ref.UNet(in_channels=2, num_classes=2, use_se=True)
        load_closed_form(net)
        net.train(training)
        x = cf.make_input(shape)
        tgt = cf.make_target((shape[0], shape[2], shape[3]), ignore_every=13)
        logits = net(x)
        loss = crit(logits, tgt)
        summarize(f"{tag}/logits", logits, g3)
        g3[f"{tag}/loss"] = np.array(loss.item())
        g3[f"{tag}/metrics"] = np.array([ref.compute_metrics(logits.detach(), tgt, 2)[k]
                                         for k in ("acc", "miou", "mpa", "mf1")])
        if training:
            loss.backward()
            for name, p in net.named_parameters():
                summarize(f"{tag}/grad/{name}", p.grad, g3)
            for name, b in net.named_buffers():
                if not name.endswith("num_batches_tracked"):
                    summarize(f"{tag}/buf/{name}", b, g3)
    # plain U-Net (use_se=False) falls out of the same classes (Unet.py equivalent)
    net = ref.UNet(in_channels=2, num_classes=2, use_se=False)
    load_closed_form(net)
    net.train(True)
    x = cf.make_input((2, 2, 32, 32))
    tgt = cf.make_target((2, 32, 32))
    logits = net(x)
    loss = crit(logits, tgt)
    loss.backward()
    summarize("nose_b2_32_train/logits", logits, g3)
    g3["nose_b2_32_train/loss"] = np.array(loss.item())
    for name, p in net.named_parameters():
        summarize(f"nose_b2_32_train/grad/{name}", p.grad, g3)
    g3["state_dict_keys"] = np.array(list(ref.UNet(2, 2, True).state_dict().keys()))
    g3["state_dict_shapes"] = np.array([str(tuple(v.shape)) for v in ref.UNet(2, 2, True).state_dict().values()])
    np.savez_compressed(os.path.join(OUT, "g3_unet.npz"), **g3)

    # ---- G3r / G4r: generic-position (PCG64) weights and inputs: well-conditioned gradients ---------------
    g3r = {}
    net = ref.UNet(in_channels=2, num_classes=2, use_se=True)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
    net.train(True)
    x = cf.make_input_random((2, 2, 64, 64), seed=11)
    tgt = cf.make_target_random((2, 64, 64), seed=13, ignore_frac=0.05)
    logits = net(x)
    loss = crit(logits, tgt)
    loss.backward()
    summarize("b2_64_train/logits", logits, g3r)
    g3r["b2_64_train/loss"] = np.array(loss.item())
    for name, p in net.named_parameters():
        summarize(f"b2_64_train/grad/{name}", p.grad, g3r)
    for name, b in net.named_buffers():
        if not name.endswith("num_batches_tracked"):
            summarize(f"b2_64_train/buf/{name}", b, g3r)
    net = ref.UNet(in_channels=2, num_classes=2, use_se=True)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
    net.train(True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    start = {k: v.clone() for k, v in net.state_dict().items()}
    losses = []
    for step in range(5):
        x = cf.make_input_random((2, 2, 64, 64), seed=100 + step)
        tgt = cf.make_target_random((2, 64, 64), seed=200 + step)
        opt.zero_grad()
        loss = crit(net(x), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    g3r["adam/losses"] = np.array(losses)
    for k, v in net.state_dict().items():
        if v.dtype == torch.float32:
            summarize(f"adam/delta/{k}", v - start[k], g3r)
            summarize(f"adam/final/{k}", v, g3r)
    np.savez_compressed(os.path.join(OUT, "g3r_unet_random.npz"), **g3r)

    # ---- G4: five Adam steps -------------------------------------------------------------
    g4 = {}
    net = ref.UNet(in_channels=2, num_classes=2, use_se=True)
    load_closed_form(net)
    net.train(True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    start = {k: v.clone() for k, v in net.state_dict().items()}
    losses = []
    for step in range(5):
        x = cf.make_input((2, 2, 64, 64), salt=0.37 * step)
        tgt = cf.make_target((2, 64, 64))
        opt.zero_grad()
        loss = crit(net(x), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    g4["losses"] = np.array(losses)
    for k, v in net.state_dict().items():
        if v.dtype == torch.float32:
            summarize(f"delta/{k}", v - start[k], g4)
            summarize(f"final/{k}", v, g4)
    np.savez_compressed(os.path.join(OUT, "g4_adam.npz"), **g4)

    # ---- G5: cross entropy ------------------------------------------------------------------
    g5 = {}
    lg = (cf.make_input((2, 2, 16, 16), 0.9) * 3.0).requires_grad_(True)
    tgt = cf.make_target((2, 16, 16), ignore_every=5)
    loss = crit(lg, tgt)
    loss.backward()
    g5["loss"] = np.array(loss.item())
    g5["dlogits"] = lg.grad.numpy()
    lg3 = (cf.make_input((2, 3, 8, 8), 0.1) * 2.0).requires_grad_(True)
    tgt3 = (cf.make_target((2, 8, 8)) + cf.make_target((2, 8, 8), 0).roll(1, 2)).clamp(max=2)
    loss3 = nn.CrossEntropyLoss(ignore_index=255)(lg3, tgt3)
    loss3.backward()
    g5["loss3"] = np.array(loss3.item())
    g5["dlogits3"] = lg3.grad.numpy()
    g5["target3"] = tgt3.numpy()
    np.savez_compressed(os.path.join(OUT, "g5_ce.npz"), **g5)

    # ---- G6: compute_metrics known-answer tests -----------------------
    g6 = {}
    def logits_for(pred):
        p = torch.tensor(pred)
        return torch.stack([(p == 0).float(), (p == 1).float()], 0).unsqueeze(0)
    cases = {
        "three_of_four": (logits_for([[0, 1], [1, 1]]), torch.tensor([[[0, 1], [0, 1]]])),
        "all_tie": (torch.zeros(1, 2, 2, 2), torch.tensor([[[0, 0], [1, 1]]])),
        "class1_absent": (logits_for([[0, 0], [0, 0]]), torch.tensor([[[0, 0], [0, 0]]])),
        "ignore255": (logits_for([[0, 1], [1, 0]]), torch.tensor([[[0, 1], [255, 255]]])),
    }
    for name, (lg_, m_) in cases.items():
        r = ref.compute_metrics(lg_, m_, 2)
        g6[f"{name}/logits"] = lg_.numpy()
        g6[f"{name}/mask"] = m_.numpy()
        g6[f"{name}/expect"] = np.array([r["acc"], r["miou"], r["mpa"], r["mf1"]], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g6_metrics.npz"), **g6)

    # ---- G9: data-parallel oracle: mean of per-shard gradients (local BN) -------------------
    g9 = {}
    grads = []
    for r in range(2):
        net = ref.UNet(in_channels=2, num_classes=2, use_se=True)
        load_closed_form(net)
        net.train(True)
        x = cf.make_input((2, 2, 32, 32), salt=1.1 * r)
        tgt = cf.make_target((2, 32, 32))
        crit(net(x), tgt).backward()
        grads.append({n: p.grad.clone() for n, p in net.named_parameters()})
    for n in grads[0]:
        summarize(f"mean_grad/{n}", 0.5 * (grads[0][n] + grads[1][n]), g9)
    np.savez_compressed(os.path.join(OUT, "g9_dp.npz"), **g9)

    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote fixtures to {OUT}: {total/1e6:.2f} MB")


if __name__ == "__main__":
    sys.exit(main())
