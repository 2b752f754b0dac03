import os, sys, torch, torch.nn.functional as F
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
torch.set_num_threads(16)
import insar_unet_ca_amd as iu
from oracle import closed_form as cf, unet_ca_oracle as orc
dev = torch.device("cuda:0")
def rl2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())
cin, cout, se = 128, 64, True
mod = iu.DoubleConv(cin, cout, use_se=se)
mod.load_state_dict(cf.fill_state_dict_random(mod.state_dict(), seed=3))
sd = {k: v.clone().double() for k, v in mod.state_dict().items()}
mod = mod.to(dev).train()
x0 = cf.make_input_random((2, cin, 64, 64), seed=5)
x = x0.to(dev).requires_grad_(True)
out = mod(x)
g = cf.make_input_random(tuple(out.shape), seed=6) * float(os.environ.get("GSCALE", "0.1"))
out.backward(g.to(dev)); torch.cuda.synchronize()
runner = list(mod._plans.plans.values())[0][0]
u1, u2, sest = runner.plan.u1, runner.plan.u2, runner.plan.se
grads = {}
def keep(n):
    def h(gr): grads[n] = gr.clone()
    return h
p = "double_conv"
xr = x0.double().requires_grad_(True)
y1 = F.conv2d(xr, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], padding=1); y1.register_hook(keep("dy1"))
z1 = torch.relu(F.batch_norm(y1, None, None, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"], training=True, eps=1e-5)); z1.register_hook(keep("dz1"))
y2 = F.conv2d(z1, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"], padding=1); y2.register_hook(keep("dy2"))
z2 = torch.relu(F.batch_norm(y2, None, None, sd[f"{p}.4.weight"], sd[f"{p}.4.bias"], training=True, eps=1e-5)); z2.register_hook(keep("dz2"))
w1, w2 = sd[f"{p}.6.fc.0.weight"], sd[f"{p}.6.fc.2.weight"]
o = orc.se_layer(z2, w1, w2)
o.backward(g.double())
print("out", rl2(out, o), " dy2", rl2(u2.dy.nchw(), grads["dy2"]), " dz1", rl2(runner.plan.dz1.nchw(), grads["dz1"]), " dy1", rl2(u1.dy.nchw(), grads["dy1"]), " dx", rl2(x.grad, xr.grad))
# terms
N = 2 * 64 * 64
y2raw = (y2.detach() - sd[f"{p}.3.bias"].view(1, -1, 1, 1))
mask = (z2 > 0).double()
gate_ref = torch.sigmoid(torch.relu(z2.detach().mean((2, 3)) @ w1.t()) @ w2.t())
print("y2raw", rl2(u2.y.nchw(), y2raw), " gate", rl2(sest.gate, gate_ref), " scale", rl2(u2.scale, sd[f"{p}.4.weight"] / torch.sqrt(y2raw.var((0,2,3), unbiased=False) + 1e-5)), " mean", rl2(u2.mean, y2raw.mean((0,2,3))))
geff_ref = grads["dz2"] * mask                      # gradient wrt BN2 output after the ReLU mask
dbeta_ref = geff_ref.sum((0, 2, 3)); 
mean = y2raw.mean((0,2,3)); istd = 1 / torch.sqrt(y2raw.var((0,2,3), unbiased=False) + 1e-5)
xh = (y2raw - mean.view(1,-1,1,1)) * istd.view(1,-1,1,1)
dgamma_ref = (geff_ref * xh).sum((0, 2, 3))
print("k1", rl2(u2.k1, dbeta_ref / N), " k2", rl2(u2.k2, dgamma_ref / N))
coefB_ref = (grads["dz2"] - g.double() * gate_ref.view(2, -1, 1, 1))   # = dsq/HW broadcast (before mask)
print("coefB", rl2(sest.coefB, coefB_ref[:, :, 0, 0]), " coefB magnitude vs dout*gate:", float(coefB_ref.abs().mean()), float((g.double()*gate_ref.view(2,-1,1,1)).abs().mean()))
# rebuild dy2 from OUR coefficients in fp64
yk = u2.y.nchw().double().cpu(); sc = u2.scale.double().cpu().view(1,-1,1,1); sh = u2.shift.double().cpu().view(1,-1,1,1)
mk = ((yk * sc + sh) > 0).double()
ge = (g.double() * sest.gate.double().cpu().view(2,-1,1,1) + sest.coefB.double().cpu().view(2,-1,1,1)) * mk
xh_o = (yk - u2.mean.double().cpu().view(1,-1,1,1)) * u2.invstd.double().cpu().view(1,-1,1,1)
dy2_rebuilt = sc * (ge - u2.k1.double().cpu().view(1,-1,1,1) - xh_o * u2.k2.double().cpu().view(1,-1,1,1))
print("dy2 rebuilt(fp64 from our coefs) vs ref", rl2(dy2_rebuilt, grads["dy2"]), " kernel dy2 vs rebuilt", rl2(u2.dy.nchw(), dy2_rebuilt))
print("mask mismatches", int((mk != mask).sum()))
