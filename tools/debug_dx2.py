import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd import engine, _lib
from oracle import closed_form as cf, unet_ca_oracle as orc
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_check import rel, halo_abs, DEV

for dtype, cin, cout, se, training in ((torch.float32, 128, 64, False, True), (torch.bfloat16, 64, 128, True, False), (torch.float32, 64, 128, True, True)):
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
    # oracle with taps on intermediate grads
    grads = {}
    def keep(name):
        def h(gr): grads[name] = gr.clone()
        return h
    xr = x0.clone().requires_grad_(True)
    p = "double_conv"
    y1 = F.conv2d(xr, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], padding=1); y1.register_hook(keep("dy1"))
    z1 = torch.relu(F.batch_norm(y1, sd[f"{p}.1.running_mean"].clone(), sd[f"{p}.1.running_var"].clone(), sd[f"{p}.1.weight"], sd[f"{p}.1.bias"], training=training, momentum=0.1, eps=1e-5)); z1.register_hook(keep("dz1"))
    y2 = F.conv2d(z1, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"], padding=1); y2.register_hook(keep("dy2"))
    z2 = torch.relu(F.batch_norm(y2, sd[f"{p}.4.running_mean"].clone(), sd[f"{p}.4.running_var"].clone(), sd[f"{p}.4.weight"], sd[f"{p}.4.bias"], training=training, momentum=0.1, eps=1e-5))
    o = orc.se_layer(z2, sd[f"{p}.6.fc.0.weight"], sd[f"{p}.6.fc.2.weight"]) if se else z2
    o.backward(g)
    print(f"== {dtype} {cin}->{cout} se={se} train={training}")
    print("  out:", rel(out, o))
    print("  dy2:", rel(u2.dy.nchw(), grads["dy2"]), " dz1:", rel(runner.plan.dz1.nchw(), grads["dz1"]), " dy1:", rel(u1.dy.nchw(), grads["dy1"]), " dx:", rel(x.grad, xr.grad))
    e = (u1.dy.nchw().cpu() - grads["dy1"])
    print("  dy1 err per-channel mean (first 6):", e.mean((0,2,3))[:6].tolist(), " err std:", e.std().item(), " dy1 true absmax", grads["dy1"].abs().max().item())
    print("  k1:", u1.k1[:4].tolist(), " k2:", u1.k2[:4].tolist())
    bn = sd[f"{p}.1.weight"]
    # expected k1 = dbeta/N
    N = 2*16*16
    gt = (grads["dz1"] * (z1 > 0)).detach()
    print("  exp k1:", (gt.sum((0,2,3))/N)[:4].tolist())
