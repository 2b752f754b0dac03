import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import insar_unet_ca_amd as iu
from insar_unet_ca_amd import engine
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")
def run(flat2, size, B):
    engine.FLAT2 = flat2
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.DiceCELoss(ignore_index=255); opt = iu.Adam(net.parameters(), lr=1e-4)
    x, y = make_batch(0, B, size); x, y = x.to(dev), y.to(dev)
    losses = []
    for it in range(6):
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), y); loss.backward(); opt.step()
        losses.append(float(loss))
    return losses
for size, B in ((96, 16), (160, 8), (256, 3), (512, 2)):
    a = run(1, size, B); b = run(0, size, B)
    print(size, B, ["%.5f" % v for v in a], ["%.5f" % v for v in b], "max rel diff %.2e" % max(abs(p - q) / abs(q) for p, q in zip(a, b)))
