import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd import engine, _lib
from oracle import closed_form as cf, unet_ca_oracle as orc
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_check import rel, halo_abs, DEV

dtype, cin, cout, se, training = torch.float32, 128, 64, False, True
mod = iu.DoubleConv(cin, cout, use_se=se)
mod.load_state_dict(cf.fill_state_dict(mod.state_dict()))
sd = {k: v.clone() for k, v in mod.state_dict().items()}
mod = mod.to(DEV); mod.compute_dtype = dtype; mod.train(training)
x0 = cf.make_input((2, cin, 16, 16))
x = x0.to(DEV).requires_grad_(True)
out = mod(x)
g = cf.make_grad(out.shape)
out.backward(g.to(DEV))
torch.cuda.synchronize()
runner = list(mod._plans.plans.values())[0][0]
u1, u2 = runner.plan.u1, runner.plan.u2
grads = {}
def keep(name):
    def h(gr): grads[name] = gr.clone()
    return h
xr = x0.clone().requires_grad_(True)
p = "double_conv"
y1 = F.conv2d(xr, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], padding=1); y1.register_hook(keep("dy1"))
z1 = torch.relu(F.batch_norm(y1, None, None, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"], training=True, eps=1e-5)); z1.register_hook(keep("dz1"))
y2 = F.conv2d(z1, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"], padding=1); y2.register_hook(keep("dy2"))
z2 = torch.relu(F.batch_norm(y2, None, None, sd[f"{p}.4.weight"], sd[f"{p}.4.bias"], training=True, eps=1e-5))
z2.backward(g)
N = 2*16*16
mask = (z2 > 0).float()
gm = g * mask
y2raw = y2.detach() - sd[f"{p}.3.bias"].view(1,-1,1,1)
print("y2 raw:", rel(u2.y.nchw(), y2raw))
mk = ((u2.y.nchw() * u2.scale.view(1,-1,1,1) + u2.shift.view(1,-1,1,1)) > 0).float().cpu()
print("mask mismatches:", int((mk != mask).sum()), "of", mask.numel())
P2 = gm.sum((2,3)); Q = (gm * y2raw).sum((2,3))
print("red P2:", rel(u2.red[:,0,:], P2), " red Q:", rel(u2.red[:,1,:], Q))
part = u2.red_part.view(2,16,2,cout)
print("red_part P2 rows:", rel(part[:,:,0,:], gm.sum(3).permute(0,2,1)))
dbeta = gm.sum((0,2,3)); mean = y2raw.mean((0,2,3)); var = y2raw.var((0,2,3), unbiased=False); istd = 1/torch.sqrt(var+1e-5)
xh = (y2raw - mean.view(1,-1,1,1))*istd.view(1,-1,1,1)
dgamma = (gm*xh).sum((0,2,3))
print("mean:", rel(u2.mean, mean), " invstd:", rel(u2.invstd, istd))
print("k1:", rel(u2.k1, dbeta/N), " k2:", rel(u2.k2, dgamma/N))
print("dgamma param grad:", rel(mod.double_conv[4].weight.grad, dgamma), " dbeta:", rel(mod.double_conv[4].bias.grad, dbeta))
dy2 = (sd[f"{p}.4.weight"]*istd).view(1,-1,1,1) * (gm - (dbeta/N).view(1,-1,1,1) - xh*(dgamma/N).view(1,-1,1,1))
print("manual dy2 vs autograd:", rel(dy2, grads["dy2"]))
e = (u2.dy.nchw().cpu() - grads["dy2"])
print("dy2 err: max", e.abs().max().item(), " relL2", (e.norm()/grads['dy2'].norm()).item(), " frac>1e-3:", (e.abs() > 1e-3*grads['dy2'].abs().max()).float().mean().item())
idx = e.abs().flatten().topk(5).indices
print("worst idx", [tuple(int(v) for v in torch.unravel_index(i, e.shape)) for i in idx])
print("dout buf vs g:", rel(runner.dout.nchw(), g))
