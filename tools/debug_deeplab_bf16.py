"""Where does the bf16 DeepLabV3-CA forward leave the fp32 one? Per-stage relative L2 error of the activations
(both on the HIP path, same weights and inputs). Usage: python tools/debug_deeplab_bf16.py B SIZE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch

B, S = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x, y = make_batch(0, B, S, channels=1)
x = x.to(dev)
acts = {}
for dt in (torch.float32, torch.bfloat16):
    torch.manual_seed(41)
    net = iu.DeepLabV3_SingleChannel_Attn(2, "resnet50", False, compute_dtype=dt)
    net.aspp.project[3].p = 0.0
    net = net.to(dev).train()
    logits = net(x)
    plan = next(iter(net._plans.plans.values()))[0]
    rec = {"y0": plan.y0.nchw(), "z0": plan.z0.nchw(), "p0": plan.p0.nchw()}
    for blk in plan.blocks:
        rec[blk.name + ".u1.y"] = blk.u1.y.nchw()
        rec[blk.name + ".u2.y"] = blk.u2.y.nchw()
        rec[blk.name + ".u3.y"] = blk.u3.y.nchw()
        rec[blk.name + ".out"] = blk.out.nchw()
    for i, u in enumerate(plan.branches):
        rec[f"aspp.{i}.y"] = u.y.nchw()
    rec["gp"] = plan.gp.nchw(); rec["pool.y"] = plan.pool_unit.y.nchw(); rec["pool.out"] = plan.pool_unit.out.nchw()
    rec["cat"] = plan.cat.nchw(); rec["project.out"] = plan.project.out.nchw(); rec["head.out"] = plan.head.out.nchw()
    rec["zc"] = plan.zc.nchw(); rec["logits_lo"] = plan.logits_lo.clone(); rec["logits"] = logits.detach().clone()
    acts[dt] = rec
    net._plans.clear(); del net, plan
    torch.cuda.empty_cache()
a, b = acts[torch.float32], acts[torch.bfloat16]
for k in a:
    den = float(a[k].norm())
    print(f"{k:28s} rel-L2 {float((a[k] - b[k]).norm()) / max(den, 1e-30):.3e}   max-rel {float((a[k]-b[k]).abs().max()/a[k].abs().max()):.3e}")
