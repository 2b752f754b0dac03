"""Multi-step race screen: RUNS repeats of a STEPS-step training run (Adam included) from the same seed must end in
one (loss, parameter checksum). GPU box only."""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")
bs = [tuple(t.to(dev) for t in make_batch(b * 16, 16, 256)) for b in range(2)]


def run(steps):
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.DiceCELoss(ignore_index=255); opt = iu.Adam(net.parameters(), lr=1e-4)
    for i in range(steps):
        x, y = bs[i % 2]
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), y); loss.backward(); opt.step()
    torch.cuda.synchronize()
    return (float(loss.detach()), float(sum(p.detach().double().sum() for p in net.parameters())))


steps, runs = int(os.environ.get("STEPS", "24")), int(os.environ.get("RUNS", "40"))
c = collections.Counter(run(steps) for _ in range(runs))
print({k: os.environ.get(k) for k in ("INSAR_SIDE_STREAM", "INSAR_C64")}, f"{runs} runs of {steps} steps: distinct results:", len(c),
      sorted(c.values(), reverse=True))
