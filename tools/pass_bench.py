"""Stand-alone timing of the HBM-bound passes of the U-Net-CA step at the five level shapes of config 2 (B = 16):
BN/ReLU apply (+pool), SE squeeze, BatchNorm-backward reduce / apply (+pool variants), Adam. HIP events around `reps`
back-to-back launches on the current stream; GB/s from the algorithmic bytes of the pass.
usage: python tools/pass_bench.py [--reps 20] [--lib path/to/variant.so ...]   (INSAR_HIP_LIB selects the library of a run;
with --lib the tool re-runs itself once per library, the default one first and last, and prints the runs side by side)"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)

LEVELS = [(256, 64), (128, 128), (64, 256), (32, 512), (16, 1024)]
if os.environ.get("PASS_BENCH_SHAPES") == "deeplab":       # the residual blocks' outputs of config 5 (DeepLabV3-CA at 256 x 256, B = 16)
    LEVELS = [(64, 256), (32, 512), (32, 1024), (32, 2048)]


def one_run(reps):
    import torch
    from insar_unet_ca_amd import engine, _lib, optim
    from insar_unet_ca_amd._lib import call, ptr
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    B = 16
    s = _lib.stream_ptr()
    out = {}

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for hw, c in LEVELS:
        A = lambda ch=c, h=hw: engine.Act.alloc(B, h, h, ch, dt, dev)
        y, g, dy, z = A(), A(), A(), A()
        y.buf[:, 1:-1, 1:-1].normal_(); g.buf[:, 1:-1, 1:-1].normal_()
        pooled, dpool = A(c, hw // 2), A(c, hw // 2)
        dpool.buf[:, 1:-1, 1:-1].normal_()
        arg = torch.zeros(B, hw // 2, hw // 2, c, dtype=torch.uint8, device=dev)
        f = lambda *sh: torch.randn(*sh, device=dev)
        scale, shift, mean, invstd, k1, k2 = f(c).abs() + 0.5, f(c) * 0.1, f(c) * 0.1, f(c).abs() + 0.5, f(c) * 1e-3, f(c) * 1e-3
        gate, coefB = torch.sigmoid(f(B, c)), f(B, c) * 1e-3
        rpp = engine._rows_per_part(B, hw)
        part = torch.zeros(B * (-(-hw // rpp)), 2, c, device=dev)
        nbytes = B * hw * hw * c * 2
        tag = f"{hw}^2 x{c}"
        runs = {
            "bn_relu_apply": (lambda: call("insar_bn_relu_apply", y.ref, ptr(scale), ptr(shift), ptr(gate), z.ref, 1, s), 2.0),
            "bn_relu_apply_pool_arg": (lambda: call("insar_bn_relu_apply_pool_arg", y.ref, ptr(scale), ptr(shift), ptr(gate), z.ref, pooled.ref, ptr(arg), 1, s), 2.0 + 0.25 + 0.125),
            "se_squeeze": (lambda: call("insar_se_squeeze", y.ref, ptr(scale), ptr(shift), ptr(part), 1, rpp, s), 1.0),
            "bwd_reduce": (lambda: call("insar_bnrelu_bwd_reduce", g.ref, y.ref, ptr(scale), ptr(shift), ptr(part), 1, rpp, s), 2.0),
            "bwd_reduce_pool": (lambda: call("insar_bnrelu_bwd_reduce_pool", g.ref, dpool.ref, ptr(arg), y.ref, ptr(scale), ptr(shift), ptr(part), 1, rpp, s), 2.0 + 0.25 + 0.125),
            "bwd_apply": (lambda: call("insar_bnrelu_bwd_apply", g.ref, y.ref, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(gate), ptr(coefB), ptr(k1), ptr(k2), dy.ref, 1, s), 3.0),
            "bwd_apply_pool": (lambda: call("insar_bnrelu_bwd_apply_pool", g.ref, dpool.ref, ptr(arg), y.ref, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(gate), ptr(coefB), ptr(k1), ptr(k2), dy.ref, 1, s), 3.0 + 0.25 + 0.125),
        }
        call("insar_bn_relu_apply_pool_arg", y.ref, ptr(scale), ptr(shift), ptr(gate), z.ref, pooled.ref, ptr(arg), 1, s)   # a valid arg-max map
        for name, (fn, passes) in runs.items():
            us = timed(fn)
            out[f"{name} {tag}"] = (us, passes * nbytes / us / 1e6)
        del y, g, dy, z, pooled, dpool
    # Adam over the parameter count of the U-Net-CA (31.04 M) in one flat buffer
    n = 31_040_000
    p = torch.nn.Parameter(torch.randn(n, device=dev))
    p.grad = torch.randn(n, device=dev) * 1e-3
    for chunk in (65536, 32768, 16384, 8192, 4096):      # elements per work-group (optim.CHUNK)
        optim.CHUNK = chunk
        opt = optim.Adam([p], lr=1e-4)
        opt.step()
        us = timed(opt.step)
        out[f"adam 31.04M chunk {chunk}"] = (us, 28.0 * n / us / 1e6)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child or not a.lib:
        for k, (us, tbs) in one_run(a.reps).items():
            print(f"{k}\t{us:.2f}\t{tbs:.3f}", flush=True)
        return
    libs = [None] + a.lib + [None]
    cols = []
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["INSAR_HIP_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--reps", str(a.reps)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode:
            sys.exit(f"run with {lib} failed:\n{r.stderr[-2000:]}")
        cols.append({l.split("\t")[0]: l.split("\t")[1:] for l in r.stdout.strip().splitlines()})
    names = ["default"] + [os.path.basename(l) for l in a.lib] + ["default again"]
    print("pass".ljust(34) + "".join(n[:24].rjust(26) for n in names))
    for k in cols[0]:
        print(k.ljust(34) + "".join(f"{float(c[k][0]):9.1f} us {float(c[k][1]):6.2f} TB/s".rjust(26) for c in cols))


if __name__ == "__main__":
    main()
