import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")
for dtype, B, S in ((torch.bfloat16, 16, 256), (torch.bfloat16, 4, 128), (torch.float32, 16, 256)):
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True, compute_dtype=dtype).to(dev).train()
    x, y = make_batch(0, B, S); x, y = x.to(dev), y.to(dev)
    crit = iu.CrossEntropyLoss(ignore_index=255)
    plan = net._plan(x)
    def snapshot():
        out = {}
        for nm, blk in [(f"enc{l}", plan.enc[l]) for l in range(5)] + [(f"dec{i}", plan.dconv[i]) for i in range(4)]:
            for un, u in (("u1", blk.u1), ("u2", blk.u2)):
                out[f"{nm}.{un}.y"] = u.y.buf.clone(); out[f"{nm}.{un}.stats"] = u.stats.clone(); out[f"{nm}.{un}.sums"] = u.sums.clone()
                out[f"{nm}.{un}.scale"] = u.scale.clone()
                if u.dy is not None: out[f"{nm}.{un}.dy"] = u.dy.buf.clone()
                out[f"{nm}.{un}.red"] = u.red.clone(); out[f"{nm}.{un}.k1"] = u.k1.clone()
            out[f"{nm}.z1"] = blk.z1.buf.clone(); out[f"{nm}.out"] = blk.out.buf.clone()
            if blk.se: out[f"{nm}.gate"] = blk.se.gate.clone(); out[f"{nm}.pooled"] = blk.se.pooled.clone()
            if blk.dz1 is not None: out[f"{nm}.dz1"] = blk.dz1.buf.clone()
        for i, a in enumerate(plan.dcat): out[f"dcat{i}"] = a.buf.clone()
        for i, a in enumerate(plan.ddec): out[f"ddec{i}"] = a.buf.clone()
        out["flatgrad"] = plan.sink.flat().clone()
        return out
    snaps = []
    for it in range(3):
        for p in net.parameters(): p.grad = None
        lg = net(x); loss = crit(lg, y); loss.backward(); torch.cuda.synchronize()
        s = snapshot(); s["logits"] = lg.detach().clone(); snaps.append(s)
    print("==", dtype, B, S)
    for a, b in ((0, 1), (1, 2)):
        bad = [k for k in snaps[a] if not torch.equal(snaps[a][k], snaps[b][k])]
        fwd = [k for k in bad if k.endswith((".y", ".stats", ".sums", ".scale", ".z1", ".gate", ".pooled", "logits"))]
        print(f" run{a} vs run{b}: {len(bad)} differing buffers; forward ones:", fwd[:12])
        for k in fwd[:3]:
            d = (snaps[a][k].float() - snaps[b][k].float()).abs()
            print("   ", k, "n_diff", int((d > 0).sum()), "max", d.max().item(), "shape", tuple(d.shape), "first idx", [int(v) for v in torch.nonzero(d)[0]] if (d>0).any() else None)
