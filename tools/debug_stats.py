import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
x, y = make_batch(0, 16, 256); x = x.to(dev)
plan = net._plan(x)
runs = []
with torch.no_grad():
    for it in range(3):
        net(x); torch.cuda.synchronize()
        u = plan.enc[1].u1
        runs.append((u.stats.clone(), u.y.nchw()))
st0, y0 = runs[0]; st1, _ = runs[1]
# reference stats from y: tile t covers flattened pixels [128t, 128t+128)
yy = y0.permute(0, 2, 3, 1).reshape(-1, 128).double()      # [M][C]
ref = torch.stack([yy.view(-1, 128, 128).sum(1), (yy.view(-1, 128, 128) ** 2).sum(1)], 1)   # [tiles][2][C]
for i, (st, _) in enumerate(runs):
    d = (st.double() - ref).abs()
    bad = d > 1e-3 * ref.abs().clamp(min=1.0)
    print(f"run{i}: entries off vs recomputed-from-y: {int(bad.sum())}")
    idx = torch.nonzero(bad)
    tiles = sorted(set(int(t) for t in idx[:, 0]))
    print("   tiles:", tiles[:24], "... n =", len(tiles))
    chans = sorted(set(int(c) for c in idx[:, 2]))
    print("   channels:", chans)
    for t, q, c in idx[:6].tolist():
        print(f"   tile {t} q {q} c {c}: got {st[t,q,c].item():.4f} ref {ref[t,q,c].item():.4f}  diff {st[t,q,c].item()-ref[t,q,c].item():.4f}")
