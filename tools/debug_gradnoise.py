import os, sys, torch
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
torch.set_num_threads(16)
import insar_unet_ca_amd as iu
from oracle import closed_form as cf, unet_ca_oracle as orc
dev = torch.device("cuda:0")
def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())
shape = (2, 2, 64, 64)
net = iu.UNet(2, 2, True)
net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
net = net.to(dev).train()
base = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
x = cf.make_input_random(shape, seed=11); tgt = cf.make_target_random((2, 64, 64), seed=13, ignore_frac=0.05)
lg = net(x.to(dev)); iu.CrossEntropyLoss(ignore_index=255)(lg, tgt.to(dev)).backward()
res = {}; logits = {}
for dt in (torch.float32, torch.float64):
    work, leaves = OrderedDict(), {}
    for k, v in base.items():
        t = v.to(dt).clone() if v.dtype == torch.float32 else v.clone()
        if orc.is_param(k): t.requires_grad_(True); leaves[k] = t
        work[k] = t
    l = orc.unet_forward(work, x.to(dt), True, True); logits[dt] = l.detach()
    orc.cross_entropy(l, tgt).backward()
    res[dt] = {k: v.grad.double() for k, v in leaves.items()}
print("logits: ours vs f64 %.2e, torch32 vs f64 %.2e" % (rel_l2(lg, logits[torch.float64]), rel_l2(logits[torch.float32], logits[torch.float64])))
for k, p in net.named_parameters():
    if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"): continue
    ref = res[torch.float64][k]
    print(f"{k:36s} ours {rel_l2(p.grad, ref):.2e}  torch32 {rel_l2(res[torch.float32][k], ref):.2e}")
