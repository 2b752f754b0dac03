"""G8: mIoU parity on a committed synthetic split. The same training protocol as the reference's
train_model/validate_model (Unet-ChannalAttention.py:321-399, 273-317: batch 8, CE(ignore 255), Adam,
per-batch compute_metrics averaged with sample weights, validation after every epoch) is run
  --side oracle : CPU oracle (fp32)          -> tests/golden/g8_miou_oracle.json   (build container)
  --side hip    : the HIP path (fp32 | bf16) -> gpurun_out/g8_miou_hip_<dtype>.json (GPU box)
on identical tiles in identical order (insar_unet_ca_amd.data, PCG64 seeds) from identical
closed-form initial weights."""
import argparse, json, os, sys, time
from collections import OrderedDict
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from insar_unet_ca_amd.data import make_batch
from oracle import closed_form as cf, unet_ca_oracle as orc

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", required=True, choices=["oracle", "hip"])
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--train", type=int, default=192)
    ap.add_argument("--val", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="")
    ap.add_argument("--perturb", type=float, default=0.0, help="relative input noise (oracle sensitivity study)")
    ap.add_argument("--pseed", type=int, default=1)
    ap.add_argument("--init", default="closed-form", choices=["closed-form", "random"],
                    help="initial weights: closed-form sines (G3) or PCG64 generic position (G3r, better conditioned)")
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    nb, nvb = a.train // a.batch, a.val // a.batch
    train = [make_batch(i * a.batch, a.batch, a.size) for i in range(nb)]
    val = [make_batch(i * a.batch, a.batch, a.size, heldout=True) for i in range(nvb)]
    if a.perturb:
        g = torch.Generator().manual_seed(a.pseed)
        train = [(x * (1 + a.perturb * torch.randn(x.shape, generator=g)), y) for x, y in train]
    rng = np.random.Generator(np.random.PCG64(4242))
    history = []
    t0 = time.time()
    if a.side == "oracle":
        torch.set_num_threads(min(os.cpu_count() or 1, a.threads))
        tmpl = orc.state_dict_template(2, 2, True)
        sd = cf.fill_state_dict(tmpl) if a.init == "closed-form" else cf.fill_state_dict_random(tmpl, seed=7)
        state = {}
        for ep in range(a.epochs):
            order = rng.permutation(nb)
            tl, tm = 0.0, np.zeros(4)
            for bi in order:
                x, y = train[bi]
                loss, logits = orc.train_step(sd, state, x, y, use_se=True, lr=a.lr)
                m = orc.compute_metrics(logits, y, 2)
                tl += loss * a.batch; tm += np.array([m[k] for k in ("acc", "miou", "mpa", "mf1")]) * a.batch
            vl, vm = 0.0, np.zeros(4)
            with torch.no_grad():
                for x, y in val:
                    lg = orc.unet_forward(OrderedDict(sd), x, True, False)
                    vl += float(orc.cross_entropy(lg, y)) * a.batch
                    m = orc.compute_metrics(lg, y, 2)
                    vm += np.array([m[k] for k in ("acc", "miou", "mpa", "mf1")]) * a.batch
            history.append({"epoch": ep + 1, "train_loss": tl / a.train, "train_miou": tm[1] / a.train,
                            "val_loss": vl / a.val, "val_acc": vm[0] / a.val, "val_miou": vm[1] / a.val,
                            "val_mpa": vm[2] / a.val, "val_mf1": vm[3] / a.val})
            print(json.dumps(history[-1]), f"[{time.time()-t0:.0f}s]", flush=True)
        out = a.out or os.path.join(ROOT, "tests", "golden", "g8_miou_oracle.json")
    else:
        import insar_unet_ca_amd as iu
        from insar_unet_ca_amd import _lib
        from insar_unet_ca_amd._lib import call, ptr
        dev = torch.device("cuda:0")
        dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
        net = iu.UNet(2, 2, True, compute_dtype=dt)
        net.load_state_dict(cf.fill_state_dict(net.state_dict()) if a.init == "closed-form"
                            else cf.fill_state_dict_random(net.state_dict(), seed=7))
        net = net.to(dev)
        crit = iu.CrossEntropyLoss(ignore_index=255)
        opt = iu.Adam(net.parameters(), lr=a.lr)
        train = [(x.to(dev), y.to(dev)) for x, y in train]
        val = [(x.to(dev), y.to(dev)) for x, y in val]
        counts = torch.zeros(3, 2, dtype=torch.int64, device=dev)
        def metrics(lg, y):
            lg = lg.detach().contiguous()
            call("insar_confusion", ptr(lg), ptr(y), lg.shape[0], 2, lg.shape[2] * lg.shape[3], 255, ptr(counts), _lib.stream_ptr())
            c = counts.cpu().numpy()
            m = orc.metrics_from_counts(c[0], c[1], c[2])
            return np.array([m[k] for k in ("acc", "miou", "mpa", "mf1")])
        for ep in range(a.epochs):
            order = rng.permutation(nb)
            net.train()
            tl, tm = 0.0, np.zeros(4)
            for bi in order:
                x, y = train[bi]
                opt.zero_grad()
                lg = net(x); loss = crit(lg, y); loss.backward(); opt.step()
                tl += float(loss.detach()) * a.batch; tm += metrics(lg, y) * a.batch
            net.eval()
            vl, vm = 0.0, np.zeros(4)
            with torch.no_grad():
                for x, y in val:
                    lg = net(x)
                    vl += float(crit(lg, y)) * a.batch; vm += metrics(lg, y) * a.batch
            history.append({"epoch": ep + 1, "train_loss": tl / a.train, "train_miou": tm[1] / a.train,
                            "val_loss": vl / a.val, "val_acc": vm[0] / a.val, "val_miou": vm[1] / a.val,
                            "val_mpa": vm[2] / a.val, "val_mf1": vm[3] / a.val})
            print(json.dumps(history[-1]), f"[{time.time()-t0:.0f}s]", flush=True)
        out = a.out or os.path.join(ROOT, "gpurun_out", f"g8_miou_hip_{a.dtype}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump({"config": vars(a), "history": history}, open(out, "w"), indent=1)
    print("wrote", out)

if __name__ == "__main__":
    main()
