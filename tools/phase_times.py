"""Forward / backward / optimizer split of one training step (HIP events on the main stream). GPU box only."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
crit = iu.DiceCELoss(ignore_index=255); opt = iu.Adam(net.parameters(), lr=1e-4)
x, y = make_batch(0, 16, 256); x, y = x.to(dev), y.to(dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
acc = [0.0, 0.0, 0.0, 0.0]
N = 30
for it in range(N + 5):
    e = [ev() for _ in range(5)]
    opt.zero_grad(set_to_none=True)
    e[0].record(); lg = net(x); e[1].record(); loss = crit(lg, y); e[2].record(); loss.backward(); e[3].record(); opt.step(); e[4].record()
    torch.cuda.synchronize()
    if it >= 5:
        for i in range(4): acc[i] += e[i].elapsed_time(e[i + 1])
print("side stream", os.environ.get("INSAR_SIDE_STREAM", "1"), "ms: forward %.3f  loss %.3f  backward %.3f  adam %.3f  total %.3f" % (*[a / N for a in acc], sum(acc) / N))
