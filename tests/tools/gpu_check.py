"""Bring-up harness: run every kernel family against torch/oracle references on the GPU and
print per-stage max-rel errors. Not a pytest (it keeps going after a failure so one GPU
session yields the whole picture). Usage: python tests/tools/gpu_check.py [--only name,...]"""
from __future__ import annotations

import argparse
import os
import sys
import time
import traceback
from collections import OrderedDict

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import insar_unet_ca_amd as iu  # noqa: E402
from insar_unet_ca_amd import _lib, engine  # noqa: E402
from insar_unet_ca_amd._lib import call, ptr  # noqa: E402
from oracle import closed_form as cf  # noqa: E402
from oracle import unet_ca_oracle as orc  # noqa: E402

DEV = torch.device("cuda:0")
RESULTS = []


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = b.abs().max().item()
    return float((a - b).abs().max().item() / (den if den > 0 else 1.0))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = b.norm().item()
    return float((a - b).norm().item() / (den if den > 0 else 1.0))


def report(name, err, tol):
    ok = err <= tol
    RESULTS.append((name, err, tol, ok))
    print(f"[{'ok' if ok else 'FAIL'}] {name}: max-rel {err:.3e} (tol {tol:.1e})", flush=True)


def halo_abs(a):
    t = a.buf.float().clone()
    t[:, 1:-1, 1:-1] = 0
    return float(t.abs().max())


def act_from(x, dtype):
    b, c, h, w = x.shape
    a = engine.Act.alloc(b, h, w, c, dtype, DEV)
    engine.pack_input(x.to(DEV), a)
    return a


def tols(dtype):
    # (forward max-rel, gradient rel-L2). Gradients are compared in rel-L2 because a ReLU input that
    # sits within rounding of 0 flips its mask between two fp32 implementations: a handful of
    # elements with O(1) error that max-rel would report as a failure of the whole tensor.
    return (1e-4, 1e-2) if dtype == torch.float32 else (0.15, 0.25)


# ------------------------------------------------------------------------------------------------
def check_pack(dtype):
    x = cf.make_input((2, 64, 8, 12))
    a = act_from(x, dtype)
    back = engine.unpack_output(a)
    report(f"pack/unpack {dtype}", rel(back, x), 1e-7 if dtype == torch.float32 else 4e-3)
    report(f"halo stays zero {dtype}", halo_abs(a), 0.0)
    x2 = cf.make_input((2, 2, 8, 12))
    a2 = act_from(x2, dtype)
    report(f"pack/unpack C=2 {dtype}", rel(engine.unpack_output(a2), x2), 1e-7 if dtype == torch.float32 else 4e-3)


def check_weight_prep(dtype):
    ctx = engine.Ctx(DEV, dtype)
    w = torch.nn.Parameter(cf.fill_tensor("weight", (128, 64, 3, 3), 3).to(DEV))
    gw = engine.GemmWeight(ctx, w, "conv3")
    f = gw.fwd().float()
    ref = w.detach().permute(2, 3, 0, 1).reshape(9, 128, 64)
    report(f"weight_prep conv fwd {dtype}", rel(f, ref), 1e-7 if dtype == torch.float32 else 4e-3)
    d = gw.dgrad().float()
    ref = w.detach().permute(2, 3, 1, 0).reshape(9, 64, 128)
    report(f"weight_prep conv dgrad {dtype}", rel(d, ref), 1e-7 if dtype == torch.float32 else 4e-3)
    wt = torch.nn.Parameter(cf.fill_tensor("up.weight", (128, 64, 2, 2), 5).to(DEV))
    gt = engine.GemmWeight(ctx, wt, "convT")
    ref = wt.detach().permute(2, 3, 1, 0).reshape(4, 64, 128)
    report(f"weight_prep convT fwd {dtype}", rel(gt.fwd().float(), ref), 1e-7 if dtype == torch.float32 else 4e-3)
    ref = wt.detach().permute(2, 3, 0, 1).reshape(4, 128, 64)
    report(f"weight_prep convT dgrad {dtype}", rel(gt.dgrad().float(), ref), 1e-7 if dtype == torch.float32 else 4e-3)


def check_igemm(dtype, cin, cout, shape):
    t_out, _ = tols(dtype)
    ctx = engine.Ctx(DEV, dtype)
    b, _, h, w = shape
    x = cf.make_input(shape)
    wt = cf.fill_tensor("weight", (cout, cin, 3, 3), 11)
    xa = act_from(x, dtype)
    ya = engine.Act.alloc(b, h, w, cout, dtype, DEV)
    p = torch.nn.Parameter(wt.to(DEV))
    gw = engine.GemmWeight(ctx, p, "conv3")
    rows = call("insar_igemm_num_mtiles", b * h * w, cout)
    stats = torch.zeros(rows, 2, cout, device=DEV)
    engine._igemm(xa, ya, gw.fwd(), cout, h, w, 1, engine._TAPS3, 0, stats=stats)
    torch.cuda.synchronize()
    got = ya.nchw()
    xr = xa.nchw().cpu()                       # rounded input as the kernel saw it
    wr = gw.fwd().float().cpu().reshape(3, 3, cout, cin).permute(2, 3, 0, 1)
    ref = F.conv2d(xr.double(), wr.double(), padding=1)
    report(f"igemm conv3x3 {cin}->{cout} {tuple(shape)} {dtype}", rel(got, ref), t_out)
    s = stats.sum(0).cpu()
    report(f"igemm stats sum {cin}->{cout} {dtype}", rel(s[0], got.sum((0, 2, 3)).cpu()), 1e-4)
    report(f"igemm stats sumsq {cin}->{cout} {dtype}", rel(s[1], (got.double() ** 2).sum((0, 2, 3)).cpu()), 1e-4)
    report(f"igemm halo untouched {cin}->{cout} {dtype}", halo_abs(ya), 0.0)
    # dgrad: dx = conv_transpose(dy, w)
    g = cf.make_grad((b, cout, h, w))
    ga = act_from(g, dtype)
    dxa = engine.Act.alloc(b, h, w, cin, dtype, DEV)
    engine._igemm(ga, dxa, gw.dgrad(), cin, h, w, 1, engine._TAPS3_DGRAD, 0)
    ref = F.conv_transpose2d(ga.nchw().cpu().double(), wr.double(), padding=1)
    report(f"igemm dgrad {cout}->{cin} {dtype}", rel(dxa.nchw(), ref), t_out)
    # wgrad
    gwt = torch.zeros(cout, cin, 3, 3, device=DEV)
    engine._wgrad_conv3(ctx, xa, ga, gwt)
    xg = xr.double().requires_grad_(False)
    wv = wr.double().clone().requires_grad_(True)
    F.conv2d(xg, wv, padding=1).backward(ga.nchw().cpu().double())
    report(f"wgrad {cin}->{cout} {tuple(shape)} {dtype}", rel(gwt, wv.grad), t_out)


def check_convT(dtype):
    t_out, _ = tols(dtype)
    ctx = engine.Ctx(DEV, dtype)
    mod = torch.nn.ConvTranspose2d(128, 64, 2, 2)
    mod.load_state_dict(cf.fill_state_dict(mod.state_dict()))
    mod = mod.to(DEV)
    x = cf.make_input((2, 128, 8, 8))
    xa = act_from(x, dtype)
    cat = engine.Act.alloc(2, 16, 16, 128, dtype, DEV)
    up = engine.UpPlan(ctx, mod, xa, cat.slice(64, 64), "up")
    up.forward()
    got = cat.slice(64, 64).nchw()
    wq = up.w.fwd().float().cpu().reshape(2, 2, 64, 128).permute(3, 2, 0, 1)
    ref = F.conv_transpose2d(xa.nchw().cpu().double(), wq.double(), mod.bias.detach().cpu().double(), stride=2)
    report(f"convT fwd {dtype}", rel(got, ref), t_out)
    report(f"convT fwd leaves skip half zero {dtype}", float(cat.slice(0, 64).nchw().abs().sum()), 0.0)
    g = cf.make_grad((2, 64, 16, 16))
    dcat = engine.Act.alloc(2, 16, 16, 128, dtype, DEV)
    engine.pack_input(g.to(DEV), dcat.slice(64, 64))
    sink = engine.GradSink(ctx, up.params())
    dx = engine.Act.alloc(2, 8, 8, 128, dtype, DEV)
    up.backward(dcat.slice(64, 64), sink, dx)
    gr = dcat.slice(64, 64).nchw().cpu().double()
    xv = xa.nchw().cpu().double().requires_grad_(True)
    wv = wq.double().clone().requires_grad_(True)
    bv = mod.bias.detach().cpu().double().requires_grad_(True)
    F.conv_transpose2d(xv, wv, bv, stride=2).backward(gr)
    report(f"convT dgrad {dtype}", rel(dx.nchw(), xv.grad), t_out)
    report(f"convT wgrad {dtype}", rel(sink.view(mod.weight), wv.grad), t_out)
    report(f"convT bias grad {dtype}", rel(sink.view(mod.bias), bv.grad), t_out)


def _oracle_block(mod_sd, prefix_fn, fn):
    pass


def check_se(dtype):
    t_out, t_g = tols(dtype)
    mod = iu.SELayer(64)
    mod.load_state_dict(cf.fill_state_dict(mod.state_dict()))
    mod = mod.to(DEV)
    mod.compute_dtype = dtype
    x = cf.make_input((2, 64, 8, 8))
    xg = x.to(DEV).requires_grad_(True)
    out = mod(xg)
    g = cf.make_grad(out.shape)
    out.backward(g.to(DEV))
    w1 = mod.fc[0].weight.detach().cpu().clone().requires_grad_(True)
    w2 = mod.fc[2].weight.detach().cpu().clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ro = orc.se_layer(xr, w1, w2)
    ro.backward(g)
    report(f"SELayer out {dtype}", rel(out, ro), t_out)
    report(f"SELayer dx {dtype}", rel_l2(xg.grad, xr.grad), t_g)
    report(f"SELayer dW1 {dtype}", rel_l2(mod.fc[0].weight.grad, w1.grad), t_g)
    report(f"SELayer dW2 {dtype}", rel_l2(mod.fc[2].weight.grad, w2.grad), t_g)


def check_double_conv(dtype, cin, cout, use_se, shape, training=True):
    t_out, t_g = tols(dtype)
    mod = iu.DoubleConv(cin, cout, use_se=use_se)
    mod.load_state_dict(cf.fill_state_dict(mod.state_dict()))
    sd = OrderedDict(("blk." + k, v.clone()) for k, v in mod.state_dict().items())
    mod = mod.to(DEV)
    mod.compute_dtype = dtype
    mod.train(training)
    x = cf.make_input(shape)
    need_dx = cin > 4
    xg = x.to(DEV).requires_grad_(need_dx)
    out = mod(xg)
    g = cf.make_grad(out.shape)
    out.backward(g.to(DEV))
    work, leaves = OrderedDict(sd), {}
    for k in sd:
        if orc.is_param(k):
            work[k] = sd[k].clone().requires_grad_(True)
            leaves[k] = work[k]
    xr = x.clone().requires_grad_(need_dx)
    ro = orc.double_conv(xr, work, "blk", use_se, training)
    ro.backward(g)
    tag = f"DoubleConv({cin},{cout},se={use_se},train={training}) {tuple(shape)} {dtype}"
    report(f"{tag} out", rel(out, ro), t_out)
    if need_dx:
        report(f"{tag} dx", rel_l2(xg.grad, xr.grad), t_g)
    gs = dict(mod.named_parameters())
    wscale = max(float(leaves["blk.double_conv.3.weight"].grad.abs().max()), 1e-30)
    for k, leaf in leaves.items():
        name = k[4:]
        got = gs[name].grad
        if name.endswith("double_conv.0.bias") or name.endswith("double_conv.3.bias"):
            if training:    # exactly zero on the HIP path; rounding noise in torch
                err = float((got.cpu().double() - leaf.grad.double()).abs().max()) / wscale
                report(f"{tag} grad {name} (abs/|dW|max)", err, 1e-3)
                continue
        report(f"{tag} grad {name}", rel_l2(got, leaf.grad), t_g)
    for k in sd:
        if not orc.is_param(k) and not k.endswith("tracked"):
            report(f"{tag} buf {k[4:]}", rel(mod.state_dict()[k[4:]], work[k]), t_out)


def check_unet(dtype, shape, use_se=True, training=True):
    t_out, t_g = tols(dtype)
    net = iu.UNet(2, 2, use_se=use_se)
    net.load_state_dict(cf.fill_state_dict(net.state_dict()))
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    net = net.to(DEV).set_compute_dtype(dtype)
    net.train(training)
    x = cf.make_input(shape)
    tgt = cf.make_target((shape[0], shape[2], shape[3]), ignore_every=13)
    crit = iu.CrossEntropyLoss(ignore_index=255)
    t0 = time.time()
    logits = net(x.to(DEV))
    loss = crit(logits, tgt.to(DEV))
    if training:
        loss.backward()
    torch.cuda.synchronize()
    dt = time.time() - t0
    work, leaves = OrderedDict(sd), {}
    for k in sd:
        if orc.is_param(k):
            work[k] = sd[k].clone().requires_grad_(True)
            leaves[k] = work[k]
    with torch.set_grad_enabled(training):
        ro = orc.unet_forward(work, x, use_se=use_se, training=training)
        rl = orc.cross_entropy(ro, tgt)
    tag = f"UNet(se={use_se},train={training}) {tuple(shape)} {dtype}"
    report(f"{tag} logits [{dt*1e3:.0f} ms first call]", rel(logits, ro), t_out)
    report(f"{tag} loss", abs(float(loss) - float(rl)) / abs(float(rl)), t_out)
    if dtype == torch.bfloat16:
        agree = (logits.argmax(1).cpu() == ro.argmax(1)).float().mean().item()
        report(f"{tag} argmax disagreement", 1.0 - agree, 0.025)
    if training:
        rl.backward()
        gs = dict(net.named_parameters())
        worst, worst_name = 0.0, ""
        for k, leaf in leaves.items():
            if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
                continue
            e = rel_l2(gs[k].grad, leaf.grad)
            if e > worst:
                worst, worst_name = e, k
            if e > t_g:
                report(f"{tag} grad {k}", e, t_g)
        report(f"{tag} worst param grad ({worst_name})", worst, t_g)
        wb = 0.0
        for k in sd:
            if not orc.is_param(k) and not k.endswith("tracked"):
                wb = max(wb, rel(net.state_dict()[k], work[k]))
        report(f"{tag} worst BN running stat", wb, t_out)


def check_ce_adam():
    lg = (cf.make_input((2, 2, 16, 16), 0.9) * 3.0)
    tgt = cf.make_target((2, 16, 16), ignore_every=5)
    a = lg.to(DEV).requires_grad_(True)
    loss = iu.CrossEntropyLoss(ignore_index=255)(a, tgt.to(DEV))
    loss.backward()
    b = lg.clone().requires_grad_(True)
    rl = orc.cross_entropy(b, tgt)
    rl.backward()
    report("CE loss", abs(float(loss) - float(rl)) / abs(float(rl)), 1e-6)
    report("CE dlogits", rel(a.grad, b.grad), 1e-5)
    a2 = lg.to(DEV).requires_grad_(True)
    dl = iu.DiceLoss(ignore_index=255)(a2, tgt.to(DEV))
    dl.backward()
    b2 = lg.clone().requires_grad_(True)
    rd = orc.soft_dice_loss(b2, tgt)
    rd.backward()
    report("Dice loss", abs(float(dl) - float(rd)) / abs(float(rd)), 1e-5)
    report("Dice dlogits", rel(a2.grad, b2.grad), 1e-4)
    # Adam: 3 steps on a few odd-sized tensors
    shapes = [(7,), (64, 3, 3, 3), (1000, 33), (5, 4)]
    ps = [torch.nn.Parameter(cf.fill_tensor("weight", s, i).to(DEV)) for i, s in enumerate(shapes)]
    rs = [p.detach().cpu().clone() for p in ps]
    opt = iu.Adam(ps, lr=1e-3)
    state = {}
    for step in range(3):
        gl = [cf.make_grad(s, 0.1 * step + 0.3 * i) for i, s in enumerate(shapes)]
        for p, g in zip(ps, gl):
            p.grad = g.to(DEV)
        opt.step()
        orc.adam_update(rs, gl, state, lr=1e-3)
    report("Adam 3 steps", max(rel(p, r) for p, r in zip(ps, rs)), 1e-6)
    # metrics counts
    lgm = cf.make_input((2, 2, 16, 16), 0.4)
    tg = cf.make_target((2, 16, 16), ignore_every=7)
    counts = torch.zeros(3, 2, dtype=torch.int64, device=DEV)
    lgd, tgd = lgm.to(DEV), tg.to(DEV)
    call("insar_confusion", ptr(lgd), ptr(tgd), 2, 2, 256, 255, ptr(counts), _lib.stream_ptr())
    tp, fp, fn = orc.confusion_counts(lgm, tg, 2)
    ref = torch.tensor([tp, fp, fn])
    report("confusion counts", float((counts.cpu().double() - ref).abs().max()), 0.0)


CHECKS = OrderedDict()
for _dt in (torch.float32, torch.bfloat16):
    _n = "f32" if _dt == torch.float32 else "bf16"
    CHECKS[f"pack_{_n}"] = (check_pack, (_dt,))
    CHECKS[f"wprep_{_n}"] = (check_weight_prep, (_dt,))
    CHECKS[f"igemm64_128_{_n}"] = (check_igemm, (_dt, 64, 128, (2, 64, 16, 16)))
    CHECKS[f"igemm128_64_tail_{_n}"] = (check_igemm, (_dt, 128, 64, (1, 128, 8, 24)))
    CHECKS[f"igemm256_256_{_n}"] = (check_igemm, (_dt, 256, 256, (2, 256, 8, 8)))
    CHECKS[f"convT_{_n}"] = (check_convT, (_dt,))
    CHECKS[f"se_{_n}"] = (check_se, (_dt,))
    CHECKS[f"dc_first_{_n}"] = (check_double_conv, (_dt, 2, 64, True, (2, 2, 16, 16)))
    CHECKS[f"dc_64_128_{_n}"] = (check_double_conv, (_dt, 64, 128, True, (2, 64, 16, 16)))
    CHECKS[f"dc_128_64_plain_{_n}"] = (check_double_conv, (_dt, 128, 64, False, (2, 128, 16, 16)))
    CHECKS[f"dc_eval_{_n}"] = (check_double_conv, (_dt, 64, 128, True, (2, 64, 16, 16), False))
    CHECKS[f"unet64_{_n}"] = (check_unet, (_dt, (2, 2, 64, 64)))
    CHECKS[f"unet48x80_{_n}"] = (check_unet, (_dt, (3, 2, 48, 80)))
CHECKS["unet_eval_f32"] = (check_unet, (torch.float32, (1, 2, 64, 64), True, False))
CHECKS["unet_nose_f32"] = (check_unet, (torch.float32, (2, 2, 64, 64), False, True))
CHECKS["ce_adam"] = (check_ce_adam, ())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    names = [n for n in args.only.split(",") if n] or list(CHECKS)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    for n in names:
        fn, a = CHECKS[n]
        print(f"--- {n}", flush=True)
        try:
            fn(*a)
            torch.cuda.synchronize()
        except Exception:
            traceback.print_exc()
            RESULTS.append((n, float("nan"), 0, False))
            try:
                torch.cuda.synchronize()
            except Exception:
                print("device unusable after failure; stopping", flush=True)
                break
    bad = [r for r in RESULTS if not r[3]]
    print(f"\n{len(RESULTS) - len(bad)} ok, {len(bad)} failed")
    for r in bad:
        print("  FAILED:", r[0], r[1])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
