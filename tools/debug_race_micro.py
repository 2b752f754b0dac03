"""Which stage of the 64->64 weight-gradient chain (GEMM slabs -> fold -> reduce) departs when the persistent
64->64 input-gradient kernel runs beside it on the main stream? GPU box only."""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from insar_unet_ca_amd import _lib, engine
from insar_unet_ca_amd._lib import call, ptr
dev = torch.device("cuda:0")
B, S, C = 16, 256, 64
iters = int(os.environ.get("ITERS", "3000"))
partner = os.environ.get("PARTNER", "c64")          # c64 | igemm | none
ctx = engine.Ctx(dev, torch.bfloat16)
g = torch.Generator(device="cpu").manual_seed(1)


def rand_act():
    a = engine.Act.alloc(B, S, S, C, torch.bfloat16, dev)
    a.buf[:, 1:-1, 1:-1, :] = torch.randn((B, S, S, C), generator=g).to(torch.bfloat16).to(dev)
    return a


x, dy = rand_act(), rand_act()
dz = engine.Act.alloc(B, S, S, C, torch.bfloat16, dev)
w = (torch.randn((9, C, C), generator=g) * 0.05).to(torch.bfloat16).to(dev).contiguous()
one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
tmp = engine.Act.alloc(B, S, S, C, torch.bfloat16, dev)
gw = torch.zeros((C, C, 3, 3), dtype=torch.float32, device=dev)


def csum(t):
    t = t.contiguous().view(-1)
    return int(t.view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64).sum())


res = []
for it in range(iters):
    if os.environ.get("ORDER", "side_first") == "side_first":
        with ctx.side_stream():
            engine._wgrad_conv3(ctx, x, dy, gw)
    else:      # as in the full step: the main-stream kernel is already queued when the side stream's event wait resolves
        call("insar_bn_relu_apply", x.ref, ptr(one), ptr(zero), 0, tmp.ref, 0, _lib.stream_ptr())
        with ctx.side_stream():
            engine._wgrad_conv3(ctx, x, dy, gw)
    if partner == "c64":
        engine._conv3x3_c64(dy, dz, w, 1, None)
    elif partner == "igemm":
        engine._igemm(dy, dz, w, C, S, S, 1, engine._TAPS3_DGRAD, 0)
    ctx.join_side()
    torch.cuda.synchronize()
    res.append((csum(ctx._wgrad_part),
                csum(ctx._wgrad_fold) if ctx._wgrad_fold is not None else 0, csum(gw), csum(dz.buf)))
names = ("part", "fold", "grad", "dz")
major = [collections.Counter(r[i] for r in res).most_common(1)[0][0] for i in range(4)]
bad = 0
for it, r in enumerate(res):
    d = [names[i] for i in range(4) if r[i] != major[i]]
    if d:
        bad += 1
        if bad <= 30:
            print("iter", it, "differs in", d, flush=True)
print("partner", partner, ":", bad, "of", iters, "iterations differ; part floats", ctx._wgrad_part.numel(),
      "fold floats", 0 if ctx._wgrad_fold is None else ctx._wgrad_fold.numel())
