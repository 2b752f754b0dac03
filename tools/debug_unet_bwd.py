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
shape = (2, 2, 64, 64)
net = iu.UNet(2, 2, True)
net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
net = net.to(dev).train()
base = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
x = cf.make_input_random(shape, seed=11); tgt = cf.make_target_random((2, 64, 64), seed=13, ignore_frac=0.05)
lg = net(x.to(dev)); iu.CrossEntropyLoss(ignore_index=255)(lg, tgt.to(dev)).backward(); torch.cuda.synchronize()
plan = net._plan(x.to(dev))
cap = {}
orig = orc.double_conv
def patched(xx, sd, prefix, use_se, training, eps=1e-5, momentum=0.1):
    p = f"{prefix}.double_conv"
    def keep(n):
        def h(g): cap[f"{prefix}/{n}"] = g.detach().clone()
        return h
    xx.register_hook(keep("dx")) if xx.requires_grad else None
    y1 = F.conv2d(xx, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], padding=1); y1.register_hook(keep("dy1"))
    z1 = orc._bn_relu(y1, sd, f"{p}.1", training, eps, momentum); z1.register_hook(keep("dz1"))
    y2 = F.conv2d(z1, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"], padding=1); y2.register_hook(keep("dy2"))
    z2 = orc._bn_relu(y2, sd, f"{p}.4", training, eps, momentum); z2.register_hook(keep("dz2"))
    o = orc.se_layer(z2, sd[f"{p}.6.fc.0.weight"], sd[f"{p}.6.fc.2.weight"]) if use_se else z2
    o.register_hook(keep("dout"))
    cap[f"{prefix}/y2"] = y2.detach() - sd[f"{p}.3.bias"].detach().view(1, -1, 1, 1)
    return o
orc.double_conv = patched
work, leaves = OrderedDict(), {}
for k, v in base.items():
    t = v.double().clone() if v.dtype == torch.float32 else v.clone()
    if orc.is_param(k): t.requires_grad_(True); leaves[k] = t
    work[k] = t
l = orc.unet_forward(work, x.double(), True, True)
orc.cross_entropy(l, tgt).backward()
blocks = [("conv4", plan.dconv[3], plan.ddec[0]), ("conv3", plan.dconv[2], plan.ddec[1]), ("inc", plan.enc[0], plan.dcat[0].slice(0, 64))]
for name, blk, dout in blocks:
    print(f"== {name}: dout {rl2(dout.nchw(), cap[name+'/dout']):.2e}  y2 {rl2(blk.u2.y.nchw(), cap[name+'/y2']):.2e}  dy2 {rl2(blk.u2.dy.nchw(), cap[name+'/dy2']):.2e}  dz1 {rl2(blk.dz1.nchw(), cap[name+'/dz1']):.2e}  dy1 {rl2(blk.u1.dy.nchw(), cap[name+'/dy1']):.2e}")
    e = (blk.u2.dy.nchw().double().cpu() - cap[name+'/dy2'])
    ref = cap[name+'/dy2']
    print("   dy2 err per image:", [float(e[n].norm()/ref[n].norm()) for n in range(2)], " per-channel worst:", float((e.flatten(2).norm(dim=2)/ref.flatten(2).norm(dim=2)).max()))
    pc = (e.permute(1,0,2,3).flatten(1).norm(dim=1) / ref.permute(1,0,2,3).flatten(1).norm(dim=1))
    print("   dy2 err by channel (first 8):", [f"{v:.1e}" for v in pc[:8].tolist()], " max", f"{pc.max():.1e}", "argmax", int(pc.argmax()))
    # is the error a per-channel constant (k1) or proportional to xhat (k2)?
    c = int(pc.argmax()); ec = e[:, c]; 
    print(f"   channel {c}: err mean {float(ec.mean()):.3e} err std {float(ec.std()):.3e} ref std {float(ref[:,c].std()):.3e}")
print("---- per-(n,c) coefficients of conv4")
blk = plan.dconv[3]; se = blk.se
z2 = torch.relu((cap["conv4/y2"] - cap["conv4/y2"].mean((0,2,3), keepdim=True)) / torch.sqrt(cap["conv4/y2"].var((0,2,3), unbiased=False, keepdim=True) + 1e-5) * work["conv4.double_conv.4.weight"].detach().view(1,-1,1,1) + work["conv4.double_conv.4.bias"].detach().view(1,-1,1,1))
gate_ref = torch.sigmoid(torch.relu(z2.mean((2,3)) @ work["conv4.double_conv.6.fc.0.weight"].detach().t()) @ work["conv4.double_conv.6.fc.2.weight"].detach().t())
print("gate", rl2(se.gate, gate_ref), [rl2(se.gate[n], gate_ref[n]) for n in range(2)])
mask = (z2 > 0).double()
coefB_ref = ((cap["conv4/dz2"] - cap["conv4/dout"] * gate_ref.view(2,-1,1,1)) * mask).flatten(2).sum(2) / mask.flatten(2).sum(2)
print("coefB", rl2(se.coefB, coefB_ref), [rl2(se.coefB[n], coefB_ref[n]) for n in range(2)])
d = (se.coefB.double().cpu() - coefB_ref)
print("coefB worst channels image1:", d[1].abs().topk(4).indices.tolist(), d[1].abs().topk(4).values.tolist(), "ref mag", float(coefB_ref.abs().mean()))
print("pooled cnt", rl2(se.pooled[:,0], mask.flatten(2).sum(2)), " sq", rl2(se.sq, z2.mean((2,3))))
print("---- structure of the dy2 error, conv4, image 1")
e = (blk.u2.dy.nchw().double().cpu() - cap["conv4/dy2"])[1]        # [C,H,W]
ref = cap["conv4/dy2"][1]
pc = e.flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
bad = (pc > 1e-4).nonzero().flatten().tolist()
print("bad channels:", bad)
print("their errors:", [f"{pc[c]:.1e}" for c in bad])
c = bad[0] if bad else 0
m1 = mask[1, c]; d1 = cap["conv4/dout"][1, c]
yk = cap["conv4/y2"]; xh = ((yk - yk.mean((0,2,3), keepdim=True)) / torch.sqrt(yk.var((0,2,3), unbiased=False, keepdim=True) + 1e-5))[1, c]
A = torch.stack([(d1 * m1).flatten(), m1.flatten(), torch.ones_like(m1).flatten(), xh.flatten()], 1)
sol = torch.linalg.lstsq(A, e[c].flatten().unsqueeze(1)).solution.flatten()
res = (A @ sol - e[c].flatten()).norm() / e[c].flatten().norm()
print(f"channel {c}: err ~ {sol[0]:.3e}*dout*mask + {sol[1]:.3e}*mask + {sol[2]:.3e} + {sol[3]:.3e}*xhat ; residual {res:.2e}")
print("scale", float(blk.u2.scale[c]), "gate[1,c]", float(se.gate[1, c]), "gate[0,c]", float(se.gate[0, c]), "coefB[1,c]", float(se.coefB[1, c]), "coefB[0,c]", float(se.coefB[0,c]), "k1", float(blk.u2.k1[c]), "k2", float(blk.u2.k2[c]))
# where are the errors located spatially?
ee = e[c].abs(); print("error rows with max:", ee.max(dim=1).values.topk(5).indices.tolist(), " frac of pixels with err>1e-9:", float((ee > 1e-9).float().mean()))
print("---- rebuild dy2[1,50] from OUR buffers in fp64")
u2 = blk.u2; c = 50
yk = u2.y.nchw().double().cpu()[1, c]; do = plan.ddec[0].nchw().double().cpu()[1, c]
sc, sh, mu, istd = [float(t[c]) for t in (u2.scale, u2.shift, u2.mean, u2.invstd)]
mk = (yk * sc + sh > 0).double()
ge = (do * float(se.gate[1, c]) + float(se.coefB[1, c])) * mk
reb = sc * (ge - float(u2.k1[c]) - (yk - mu) * istd * float(u2.k2[c]))
ker = u2.dy.nchw().double().cpu()[1, c]
print("kernel vs rebuilt:", float((ker - reb).norm() / reb.norm()), " rebuilt vs ref:", float((reb - cap['conv4/dy2'][1, c]).norm() / cap['conv4/dy2'][1, c].norm()))
print("ours  ", ker[0, :6].tolist()); print("reb   ", reb[0, :6].tolist()); print("ref   ", cap['conv4/dy2'][1, c][0, :6].tolist())
print("dout ours vs ref (1,50):", float((do - cap['conv4/dout'][1, c]).norm() / cap['conv4/dout'][1, c].norm()))
print("y2 ours vs ref (1,50):", float((yk - cap['conv4/y2'][1, c]).norm() / cap['conv4/y2'][1, c].norm()))
print("---- error statistics (1,50)")
er = (ker - cap['conv4/dy2'][1, c])
print("on mask: mean %.3e std %.3e | off mask: mean %.3e std %.3e | mask frac %.3f" % (float(er[mk>0].mean()), float(er[mk>0].std()), float(er[mk==0].mean()), float(er[mk==0].std()), float(mk.mean())))
xh = (yk - mu) * istd
print("corr with xhat:", float(((er-er.mean())*(xh-xh.mean())).mean()/ (er.std()*xh.std())))
for cc in (50, 55, 10):
    e0 = (u2.dy.nchw().double().cpu()[0, cc] - cap['conv4/dy2'][0, cc]); e1 = (u2.dy.nchw().double().cpu()[1, cc] - cap['conv4/dy2'][1, cc])
    print(f"channel {cc}: image0 err mean {float(e0.mean()):.3e} std {float(e0.std()):.3e}; image1 err mean {float(e1.mean()):.3e} std {float(e1.std()):.3e}")
# oracle-side sanity: does the fp32 oracle agree with the fp64 oracle on this element?
print("---- normalised regression (1,50)")
z2c = torch.relu(yk * sc + sh)
cols = {"dout*mask": do * mk, "mask": mk, "z2": z2c, "one": torch.ones_like(mk), "xhat": xh, "dout": do}
A = torch.stack([v.flatten() / v.flatten().norm() for v in cols.values()], 1)
y = er.flatten(); 
sol = torch.linalg.lstsq(A, (y / y.norm()).unsqueeze(1), rcond=1e-12).solution.flatten()
print({k: round(float(s), 4) for k, s in zip(cols, sol)}, "residual", float((A @ sol - y / y.norm()).norm()))
print("err/dout*mask ratio on mask: mean %.4e std %.4e" % (float((er[mk>0] / (do[mk>0]*sc)).mean()), float((er[mk>0] / (do[mk>0]*sc)).std())))
print("gate ours %.8f ; implied gate err" % float(se.gate[1, c]))
print("---- oracle self-check at (1,50)")
dz2_ref = cap["conv4/dz2"][1, c]; dout_ref = cap["conv4/dout"][1, c]
form = dout_ref * float(gate_ref[1, c]) + float(coefB_ref[1, c])
print("oracle dz2 vs dout*gate+coefB (all pixels): rel", float((dz2_ref - form).norm() / dz2_ref.norm()))
print("masked only:", float(((dz2_ref - form) * mk).norm() / (dz2_ref * mk).norm()), " -- note dz2 hook is BEFORE relu mask? values off-mask:", float((dz2_ref * (1 - mk)).abs().max()))
dz2_full = cap["conv4/dz2"]
# BN backward in fp64 from oracle tensors
yk_all = cap["conv4/y2"]; g_all = dz2_full * (z2 > 0)
Nn = yk_all.numel() / yk_all.shape[1]
mean_ = yk_all.mean((0,2,3), keepdim=True); istd_ = 1 / torch.sqrt(yk_all.var((0,2,3), unbiased=False, keepdim=True) + 1e-5)
xh_ = (yk_all - mean_) * istd_
gam = work["conv4.double_conv.4.weight"].detach().view(1,-1,1,1)
dy_form = gam * istd_ * (g_all - g_all.mean((0,2,3), keepdim=True) - xh_ * (g_all * xh_).mean((0,2,3), keepdim=True))
print("oracle dy2 vs BN-backward formula:", float((dy_form - cap['conv4/dy2']).norm() / cap['conv4/dy2'].norm()), " at (1,50):", float((dy_form[1,c] - cap['conv4/dy2'][1,c]).norm() / cap['conv4/dy2'][1,c].norm()))
print("ours vs formula at (1,50):", float((ker - dy_form[1, c]).norm() / dy_form[1, c].norm()))
print("---- term by term at (1,50)")
mref = (z2 > 0).double()[1, c]
print("mask mismatches:", int((mk != mref).sum()))
ge_ref = (dz2_full * (z2 > 0))[1, c]
print("ge ours vs ref:", float((ge - ge_ref).norm() / ge_ref.norm()))
k1_ref = float(g_all.mean((0,2,3))[c]); k2_ref = float((g_all * xh_).mean((0,2,3))[c])
print("k1 ours %.6e ref %.6e | k2 ours %.6e ref %.6e" % (float(u2.k1[c]), k1_ref, float(u2.k2[c]), k2_ref))
print("scale ours %.8f ref %.8f | mean ours %.8f ref %.8f | invstd ours %.8f ref %.8f" % (sc, float((gam*istd_)[0,c,0,0]), mu, float(mean_[0,c,0,0]), istd, float(istd_[0,c,0,0])))
print("xhat ours vs ref:", float(((yk - mu) * istd - xh_[1, c]).norm() / xh_[1, c].norm()))
print("x̂*k2 magnitude vs ge magnitude: %.3e vs %.3e" % (float((xh_[1,c]*k2_ref).abs().mean()), float(ge_ref.abs().mean())))
