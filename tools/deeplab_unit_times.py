"""Per-unit stand-alone times of config 5 (DeepLabV3-CA bf16, 16x1x256x256): forward conv, input-gradient GEMM(s) and weight
gradient (+ folds) of every convolution, by HIP events around the unit's calls with everything on ONE stream.
usage: python tools/deeplab_unit_times.py"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import insar_unet_ca_amd as iu
from insar_unet_ca_amd import engine, deeplab
from insar_unet_ca_amd.data import make_batch


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = iu.DeepLabV3_SingleChannel_Attn(num_classes=2, backbone="resnet50", pretrained=False, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-4)
    x, y = (t.to(dev) for t in make_batch(0, 16, 256, channels=1))
    from insar_unet_ca_amd import tape
    tape.MODE = "0"
    rec = collections.defaultdict(list)

    def wrap(cls, meth, kind):
        orig = getattr(cls, meth)

        def f(self, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(self, *a, **k)
            e1.record()
            rec[(self.name, kind)].append((e0, e1, self))
            return r
        setattr(cls, meth, f)

    wrap(deeplab.ConvUnit, "_weight_grad", "wgrad")
    os.environ["X"] = "1"
    engine.PROFILER = engine.KernelTimer(alone=True)      # single-stream mode: side-stream sections run inline
    for i in range(4):
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), y)
        loss.backward()
        opt.step()
        if i == 0:
            rec.clear()
    torch.cuda.synchronize()
    engine.PROFILER = None
    rows = []
    for (name, kind), evs in rec.items():
        u = evs[0][2]
        us = sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs) * 1e3
        fl = 2.0 * u.M * u.cin * u.cout * len(u.taps)
        rows.append((us, name, kind, u.cin, u.cout, u.k, u.s, u.d, len(u.taps), u.M, fl / us / 1e6))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"weight gradients (+ folds), alone: {tot:.0f} us over {len(rows)} units")
    for r in rows:
        print(f"  {r[0]:7.1f} us  {r[1]:34s} {r[3]:4d}->{r[4]:4d} k{r[5]} s{r[6]} d{r[7]:2d} live taps {r[8]}  M {r[9]:6d}  {r[10]:6.0f} TF")


if __name__ == "__main__":
    main()
