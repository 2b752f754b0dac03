"""Timing experiment (an upper bound for half-batch pipelining): two independent U-Net-CA training steps of 8 tiles each on two
compute streams (each plan with its own weight-gradient side stream) against one step of 16 tiles. NOT the same arithmetic (the
BatchNorm statistics are per half): only the aggregate tiles/s matters here."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")


def make(B):
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.DiceCELoss(ignore_index=255); opt = iu.Adam(net.parameters(), lr=1e-4)
    x, y = make_batch(0, B, 256); x, y = x.to(dev), y.to(dev)
    def step():
        opt.zero_grad(set_to_none=True); loss = crit(net(x), y); loss.backward(); opt.step()
    return step


def timed(steps_fn, n=30, warm=8):
    for _ in range(warm): steps_fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): steps_fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


one = make(16)
print("one step of 16 tiles: %.3f ms" % timed(one), flush=True)
a, b = make(8), make(8)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def both():
    with torch.cuda.stream(sa): a()
    with torch.cuda.stream(sb): b()
print("two steps of 8 tiles on two streams: %.3f ms per pair" % timed(both), flush=True)
print("one step of 8 tiles alone: %.3f ms" % timed(a), flush=True)
