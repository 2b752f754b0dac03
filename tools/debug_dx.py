import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd import engine, _lib
from oracle import closed_form as cf
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_check import rel, halo_abs, DEV

for dtype, cin, cout, se in ((torch.float32, 128, 64, False), (torch.bfloat16, 64, 128, True), (torch.float32, 64, 128, True)):
    mod = iu.DoubleConv(cin, cout, use_se=se)
    mod.load_state_dict(cf.fill_state_dict(mod.state_dict()))
    mod = mod.to(DEV); mod.compute_dtype = dtype; mod.train()
    x = cf.make_input((2, cin, 16, 16)).to(DEV).requires_grad_(True)
    out = mod(x)
    out.backward(cf.make_grad(out.shape).to(DEV))
    torch.cuda.synchronize()
    runner = list(mod._plans.plans.values())[0][0]
    u1, u2 = runner.plan.u1, runner.plan.u2
    print(f"== {dtype} {cin}->{cout} se={se}")
    for nm, a in (("u1.dy", u1.dy), ("u2.dy", u2.dy), ("dz1", runner.plan.dz1), ("dout", runner.dout), ("dx", runner.dx), ("z1", runner.plan.z1), ("y1", u1.y), ("y2", u2.y)):
        print(f"  halo {nm}: {halo_abs(a):.3e}  interior absmax {a.nchw().abs().max().item():.3e}")
    W = mod.double_conv[0].weight.detach()
    Wq = u1.w.dgrad().float().reshape(3, 3, cin, cout).permute(3, 2, 0, 1)   # back to (co,ci,3,3)
    print("  dgrad weights vs param:", rel(Wq, W))
    ref = F.conv_transpose2d(u1.dy.nchw().double(), Wq.double(), padding=1)
    print("  dx vs conv_transpose(dy1):", rel(runner.dx.nchw(), ref))
    dx2 = engine.Act.alloc(2, 16, 16, cin, dtype, DEV)
    engine._igemm(u1.dy, dx2, u1.w.dgrad(), cin, 16, 16, 1, engine._TAPS3_DGRAD, 0)
    torch.cuda.synchronize()
    print("  re-run dgrad vs ref:", rel(dx2.nchw(), ref), " vs first:", rel(dx2.nchw(), runner.dx.nchw()))
    print("  x.grad vs runner.dx:", rel(x.grad, runner.dx.nchw()))
