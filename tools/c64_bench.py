"""Timing experiments for the 64->64 channel kernel: per-tile vs fixed cost, and what the time is spent on."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from insar_unet_ca_amd import _lib, engine
from insar_unet_ca_amd._lib import call, ptr
dev = torch.device("cuda:0")
dtype = torch.bfloat16
ctx = engine.Ctx(dev, dtype)
p = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
gw = engine.GemmWeight(ctx, p, "conv3")
def run(B, H, W, flag, with_stats=True, iters=20):
    xa = engine.Act.alloc(B, H, W, 64, dtype, dev)
    xa.buf[:, 1:-1, 1:-1] = torch.randn(B, H, W, 64, device=dev).to(dtype)
    ya = engine.Act.alloc(B, H, W, 64, dtype, dev)
    rows = call("insar_conv3x3_c64_rows", xa.ref)
    stats = torch.zeros(rows, 2, 64, device=dev)
    for _ in range(3):
        call("insar_conv3x3_c64", xa.ref, ya.ref, ptr(gw.fwd()), flag, ptr(stats) if with_stats else 0, _lib.stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call("insar_conv3x3_c64", xa.ref, ya.ref, ptr(gw.fwd()), flag, ptr(stats) if with_stats else 0, _lib.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * B * H * W * 64 * 64 * 9
    return us, fl / us / 1e6
for B in (4, 8, 16, 32):
    us, tf = run(B, 256, 256, 0)
    print(f"B={B:3d} 256x256: {us:8.1f} us  {tf:7.1f} TF/s   tiles/WG {B*258*258/256/256:.1f}")
