"""Layer-by-layer forward error of the fp32 HIP path (and of torch's own fp32) against the float64 oracle,
G3r fixture. Prints rel-L2 of y1 (raw conv1 output, bias removed), z1, y2, block output."""
import os, sys, torch, torch.nn.functional as F
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
torch.set_num_threads(16)
import insar_unet_ca_amd as iu
from oracle import closed_form as cf, unet_ca_oracle as orc
dev = torch.device("cuda:0")
def rl2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())
shape = (2, 2, 64, 64)
net = iu.UNet(2, 2, True)
net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
net = net.to(dev).train()
base = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
x = cf.make_input_random(shape, seed=11)
with torch.no_grad():
    lg = net(x.to(dev))
torch.cuda.synchronize()
plan = net._plan(x.to(dev))
orig = orc.double_conv
def run(dt):
    cap = {}
    def patched(xx, sd, prefix, use_se, training, eps=1e-5, momentum=0.1):
        p = f"{prefix}.double_conv"
        y1 = F.conv2d(xx, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], padding=1)
        z1 = orc._bn_relu(y1, sd, f"{p}.1", training, eps, momentum)
        y2 = F.conv2d(z1, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"], padding=1)
        z2 = orc._bn_relu(y2, sd, f"{p}.4", training, eps, momentum)
        o = orc.se_layer(z2, sd[f"{p}.6.fc.0.weight"], sd[f"{p}.6.fc.2.weight"]) if use_se else z2
        cap[prefix] = dict(x=xx.detach(), y1=y1.detach() - sd[f"{p}.0.bias"].view(1, -1, 1, 1), z1=z1.detach(),
                           y2=y2.detach() - sd[f"{p}.3.bias"].view(1, -1, 1, 1), out=o.detach())
        return o
    orc.double_conv = patched
    work = OrderedDict((k, (v.to(dt).clone() if v.dtype == torch.float32 else v.clone())) for k, v in base.items())
    with torch.no_grad():
        l = orc.unet_forward(work, x.to(dt), True, True)
    orc.double_conv = orig
    return cap, l
c64, l64 = run(torch.float64)
c32, l32 = run(torch.float32)
names = ["inc", "down1.1", "down2.1", "down3.1", "down4.1"]
blocks = list(zip(names, plan.enc)) + list(zip(["conv1", "conv2", "conv3", "conv4"], plan.dconv))
print("%-9s | %-43s | %s" % ("block", "ours: x y1 z1 y2 out", "torch fp32: y1 z1 y2 out"))
for name, blk in blocks:
    r = c64[name]; t = c32[name]
    ours = (rl2(blk.x.nchw(), r["x"]), rl2(blk.u1.y.nchw(), r["y1"]), rl2(blk.z1.nchw(), r["z1"]), rl2(blk.u2.y.nchw(), r["y2"]), rl2(blk.out.nchw(), r["out"]))
    th = (rl2(t["y1"], r["y1"]), rl2(t["z1"], r["z1"]), rl2(t["y2"], r["y2"]), rl2(t["out"], r["out"]))
    print("%-9s | %s | %s" % (name, " ".join("%.1e" % v for v in ours), " ".join("%.1e" % v for v in th)))
print("---- ReLU decisions that differ from the float64 oracle, and |pre-activation| (float64, in units of the channel std) there")
def pre(cap, name, which):
    y = cap[name]["y1" if which == 1 else "y2"].double()
    k = f"{name}.double_conv.{1 if which == 1 else 4}"
    mu = y.mean((0, 2, 3), keepdim=True); var = y.var((0, 2, 3), unbiased=False, keepdim=True)
    return (y - mu) / torch.sqrt(var + 1e-5) * base[k + ".weight"].double().view(1, -1, 1, 1) + base[k + ".bias"].double().view(1, -1, 1, 1)
for name, blk in blocks:
    for which, ours_act, key in ((1, blk.z1, "z1"), (2, blk.out, "out")):
        z = pre(c64, name, which)
        m64 = z > 0
        mo = ours_act.nchw().cpu() > 0
        mt = c32[name][key] > 0
        fo = (mo != m64); ft = (mt != m64)
        zo = z[fo].abs()
        print("%-9s bn%d  ours %3d flips (|z| max %.1e)   torch32 %3d flips (|z| max %.1e)   elements with |z|<1e-5: %d" % (
            name, which, int(fo.sum()), float(zo.max()) if zo.numel() else 0.0, int(ft.sum()), float(z[ft].abs().max()) if ft.any() else 0.0, int((z.abs() < 1e-5).sum())))
print("logits ours %.2e torch %.2e" % (rl2(lg, l64), rl2(l32, l64)))
