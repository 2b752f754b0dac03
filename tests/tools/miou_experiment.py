"""mIoU-parity experiment (north_star: "mIoU within 0.3 pt of the reference on identical splits").

Both sides run the reference's OWN protocol — train_model / validate_model of Unet-ChannalAttention.py:321-399,
273-317 (batch 8, CrossEntropyLoss(ignore_index=255), Adam(lr=1e-4), per-batch compute_metrics averaged with
sample weights) — on the synthetic "bowl" task (insar_unet_ca_amd.data.make_tile_bowl), for several seeds. A seed
fixes the initial weights (oracle.closed_form.fill_state_dict_random) and the batch order (data.SeededBatches); the
tiles are the same for every seed. After training, validate_model runs once more on a large held-out set.

  --side reference : the IMPORTED reference (its UNet, train_model, validate_model, compute_metrics; torch CPU fp32),
                     build container only           -> tests/golden/g8c_miou_reference.json (committed fixture)
  --side hip       : insar_unet_ca_amd (UNet, train.train_model / validate_model, CrossEntropyLoss, Adam) on the GPU
                                                    -> gpurun_out/g8c_miou_hip_<dtype>.json
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from insar_unet_ca_amd.data import SeededBatches, make_batch   # noqa: E402
from oracle import closed_form as cf                           # noqa: E402

DEFAULTS = dict(size=64, train=256, val=32, heldout=1024, batch=8, epochs=30, lr=1e-4, task="bowl")


def build_batches(a):
    mk = lambda n, held, first=0: [make_batch(first + i * a.batch, a.batch, a.size, heldout=held, task=a.task)
                                   for i in range(n // a.batch)]
    return mk(a.train, False), mk(a.val, True), mk(a.heldout, True, first=a.val)


def run_seed(a, seed, train_b, val_b, held_b):
    t0 = time.time()
    if a.side == "reference":
        from oracle import ref_loader
        ref = ref_loader.load_reference_unet_ca()
        ref.MODEL_SAVE_PATH = os.path.join(a.tmp, f"ref_best_{seed}.pth")
        net = ref.UNet(in_channels=2, num_classes=2, use_se=True)
        net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=seed))
        crit = torch.nn.CrossEntropyLoss(ignore_index=255)
        opt = torch.optim.Adam(net.parameters(), lr=a.lr)
        dev = torch.device("cpu")
        tr, va, he = SeededBatches(train_b, True, seed), SeededBatches(val_b, False), SeededBatches(held_b, False)
        with contextlib.redirect_stdout(io.StringIO()):
            hist = ref.train_model(net, tr, va, crit, opt, dev, num_epochs=a.epochs)
            final = ref.validate_model(net, he, crit, dev)
    else:
        import insar_unet_ca_amd as iu
        from insar_unet_ca_amd import train as T
        dev = torch.device("cuda:0")
        dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
        net = iu.UNet(in_channels=2, num_classes=2, use_se=True, compute_dtype=dt)
        net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=seed))
        crit = iu.CrossEntropyLoss(ignore_index=255)
        net = net.to(dev)
        opt = iu.Adam(net.parameters(), lr=a.lr)
        tr, va, he = (SeededBatches(train_b, True, seed, dev), SeededBatches(val_b, False, device=dev),
                      SeededBatches(held_b, False, device=dev))
        hist = T.train_model(net, tr, va, crit, opt, dev, num_epochs=a.epochs, verbose=False)
        final = T.validate_model(net, he, crit, dev, verbose=False)
    rec = {"seed": seed, "final": {k: float(v) for k, v in final.items()},
           "val_miou_curve": [float(h["val_miou"]) for h in hist], "train_loss_curve": [float(h["train_loss"]) for h in hist],
           "seconds": round(time.time() - t0, 1)}
    print(json.dumps({"seed": seed, "heldout_miou": rec["final"]["val_miou"], "heldout_loss": rec["final"]["val_loss"],
                      "last_val_miou": rec["val_miou_curve"][-1], "s": rec["seconds"]}), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", required=True, choices=["reference", "hip"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--seeds", default="0,1,2,3,4")
    for k, v in DEFAULTS.items():
        ap.add_argument(f"--{k}", type=type(v), default=v)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--out", default="")
    ap.add_argument("--tmp", default="/tmp")
    a = ap.parse_args()
    if a.side == "reference":
        torch.set_num_threads(a.threads)
    train_b, val_b, held_b = build_batches(a)
    out = a.out or (os.path.join(ROOT, "tests", "golden", "g8c_miou_reference.json") if a.side == "reference"
                    else os.path.join(ROOT, "gpurun_out", f"g8c_miou_hip_{a.dtype}.json"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    runs = []
    for seed in [int(s) for s in a.seeds.split(",")]:
        runs.append(run_seed(a, seed, train_b, val_b, held_b))
        cfg = {k: getattr(a, k) for k in DEFAULTS}
        cfg.update(side=a.side, dtype=a.dtype if a.side == "hip" else "f32",
                   generator="tests/tools/miou_experiment.py", torch=torch.__version__)
        json.dump({"config": cfg, "runs": runs}, open(out, "w"), indent=1)      # rewritten after every seed
    m = np.array([r["final"]["val_miou"] for r in runs])
    print(f"held-out mIoU over {len(m)} seeds: mean {m.mean():.4f}  std {m.std(ddof=1) if len(m) > 1 else 0:.4f}  -> {out}")


if __name__ == "__main__":
    main()
