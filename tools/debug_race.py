"""Localise run-to-run differences of one full-size training step (no optimizer): repeat fwd+bwd on fixed
weights and inputs, checksum every plan buffer and every parameter gradient, and list the buffers whose
checksum departs from the majority value. GPU box only."""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd.data import make_batch
dev = torch.device("cuda:0")
B, S = int(os.environ.get("B", "16")), int(os.environ.get("S", "256"))
runs = int(os.environ.get("RUNS", "600"))
torch.manual_seed(0)
net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
x, y = make_batch(0, B, S); x, y = x.to(dev), y.to(dev)
crit = iu.DiceCELoss(ignore_index=255)
plan = net._plan(x)
from insar_unet_ca_amd import engine
SEP = os.environ.get("SEPARATE_SCRATCH", "0") == "1"
store = {}
if SEP:        # one slab / fold buffer per layer instead of the shared scratch, so that they can be inspected after the step
    _orig = engine._wgrad_conv3
    def _patched(ctx, x_, dy_, grad):
        key = grad.data_ptr()
        ctx._wgrad_part, ctx._wgrad_fold = store.get(key, (None, None))
        _orig(ctx, x_, dy_, grad)
        store[key] = (ctx._wgrad_part, ctx._wgrad_fold)
    engine._wgrad_conv3 = _patched
WATCH = ["inc.double_conv.3.weight", "conv4.double_conv.3.weight"]
names = {id(p): n for n, p in net.named_parameters()}


def csum(t):
    t = t.contiguous().view(-1)
    if t.element_size() == 2:
        return int(t.view(torch.int16).to(torch.int64).sum())
    return int(t.view(torch.int32).to(torch.int64).sum())


def snapshot(lg):
    out = collections.OrderedDict()
    out["logits"] = csum(lg)
    blocks = [(f"enc{l}", plan.enc[l]) for l in range(5)] + [(f"dec{i}", plan.dconv[i]) for i in range(4)]
    for nm, blk in blocks:
        for un, u in (("u1", blk.u1), ("u2", blk.u2)):
            out[f"{nm}.{un}.y"] = csum(u.y.buf); out[f"{nm}.{un}.stats"] = csum(u.stats)
            out[f"{nm}.{un}.scale"] = csum(u.scale)
            if u.dy is not None:         # (the first unit's dy is never written: its apply pass runs inside its weight-gradient kernel)
                out[f"{nm}.{un}.dy"] = csum(u.dy.buf)
            out[f"{nm}.{un}.k1"] = csum(u.k1); out[f"{nm}.{un}.red"] = csum(u.red_part)
        out[f"{nm}.z1"] = csum(blk.z1.buf); out[f"{nm}.out"] = csum(blk.out.buf); out[f"{nm}.dz1"] = csum(blk.dz1.buf)
        if blk.se:
            out[f"{nm}.gate"] = csum(blk.se.gate)
    for i, a in enumerate(plan.pooled): out[f"pooled{i}"] = csum(a.buf)
    for i, a in enumerate(plan.dcat): out[f"dcat{i}"] = csum(a.buf)
    for i, a in enumerate(plan.ddec): out[f"ddec{i}"] = csum(a.buf)
    for i, a in enumerate(plan.dpooled): out[f"dpooled{i}"] = csum(a.buf)
    out["dx5"] = csum(plan.dx5.buf)
    for p in plan.grad_params:
        out["grad:" + names[id(p)]] = csum(plan.sink.view(p))
        if SEP and names[id(p)] in WATCH:
            pt, fd = store[plan.sink.view(p).data_ptr()]
            out["part:" + names[id(p)]] = csum(pt)
            out["fold:" + names[id(p)]] = csum(fd) if fd is not None else 0
    return out


params = dict(net.named_parameters())
first = {}
keep = []


snaps = []
for it in range(runs):
    for p in net.parameters(): p.grad = None
    lg = net(x); loss = crit(lg, y); loss.backward(); torch.cuda.synchronize()
    snaps.append(snapshot(lg.detach()))
    for wn in WATCH:
        gcur = plan.sink.view(params[wn])
        if it == 0:
            first[wn] = gcur.clone()
        elif snaps[-1]["grad:" + wn] != snaps[0]["grad:" + wn]:
            d = (gcur - first[wn]).abs().view(-1)
            nz = torch.nonzero(d).view(-1)
            print(f"run {it} {wn}: {nz.numel()} of {d.numel()} elements differ from run 0; max |d| {float(d.max()):.3e} "
                  f"(max |g| {float(first[wn].abs().max()):.3e}); first flat idx {nz[:12].tolist()} last {nz[-3:].tolist()}", flush=True)
keys = list(snaps[0].keys())
major = {k: collections.Counter(s[k] for s in snaps).most_common(1)[0][0] for k in keys}
nbad = 0
for i, s in enumerate(snaps):
    bad = [k for k in keys if s[k] != major[k]]
    if bad:
        nbad += 1
        print(f"run {i}: {len(bad)} buffers differ:", bad[:40], flush=True)
print({k: os.environ.get(k) for k in ("INSAR_SIDE_STREAM", "INSAR_C64")}, f"{nbad} of {runs} runs differ from the majority")
