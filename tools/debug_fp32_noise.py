"""How exact is the fp32 implicit-GEMM? Compares, on one 3x3 conv (64->64 and 128->64 at 64x64, B=2):
ours (16x16x4 f32 MFMA chain), torch-ROCm fp32 conv (MIOpen), torch CPU fp32 — all against float64."""
import os, sys, torch, numpy as np
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
torch.set_num_threads(16)
from insar_unet_ca_amd import engine, _lib
from insar_unet_ca_amd._lib import call
dev = torch.device("cuda:0")
rng = np.random.Generator(np.random.PCG64(5))
for cin, cout in ((64, 64), (128, 64), (512, 512)):
    B, H, W = 2, 64, 64
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    x = torch.relu(x)        # like a post-ReLU activation (non-negative => partial sums do not cancel on the x side)
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (cin * 9)) ** 0.5).astype(np.float32))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    ctx = engine.Ctx(dev, torch.float32)
    xa = engine.Act.alloc(B, H, W, cin, torch.float32, dev); engine.pack_input(x.to(dev), xa)
    ya = engine.Act.alloc(B, H, W, cout, torch.float32, dev)
    gw = engine.GemmWeight(ctx, torch.nn.Parameter(w.to(dev)), "conv3")
    engine._igemm(xa, ya, gw.fwd(), cout, H, W, 1, engine._TAPS3, 0)
    ours = ya.nchw().double().cpu()
    tg = F.conv2d(x.to(dev), w.to(dev), padding=1).double().cpu()
    tc = F.conv2d(x, w, padding=1).double()
    def stats(a):
        e = a - ref
        return "rel-L2 %.2e  max/absmax %.2e  mean signed err/rms(ref) %+.2e" % (float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max()), float(e.mean() / ref.pow(2).mean().sqrt()))
    print(f"cin {cin} cout {cout} K {cin*9}")
    print("  ours (MFMA f32)   ", stats(ours))
    print("  torch ROCm fp32   ", stats(tg))
    print("  torch CPU fp32    ", stats(tc))
