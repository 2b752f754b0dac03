"""Where a work-group of the per-tap implicit-GEMM kernel (csrc/igemm.hip) and of the row-of-taps weight-gradient kernel
(csrc/wgrad3.hip) spends its cycles: in-kernel s_memtime stamps of wave 0 (diagnostic build -DINSAR_STAMPS ->
insar_unet_ca_amd/libinsar_hip_stamps.so; the product library has no stamps).
usage: INSAR_HIP_LIB=insar_unet_ca_amd/libinsar_hip_stamps.so python tools/stamp_gemm.py"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from insar_unet_ca_amd import engine, _lib
from insar_unet_ca_amd._lib import call, ptr

IG = ["row tables", "first slab landed", "K loop", "acc -> LDS tile", "stores (+ sums)", "statistics fold", "-", "-"]
W3 = ["set-up", "first step landed", "K loop", "slab stores", "-", "-", "-", "-"]
LAYERS = {"down2.3": (256, 256, 64), "conv2.0": (512, 256, 64), "down3.3": (512, 512, 32), "conv1.0": (1024, 512, 32),
          "down4.0": (512, 1024, 16), "down4.3": (1024, 1024, 16), "conv3.0": (256, 128, 128), "down1.3": (128, 128, 128)}


def report(lib, which, names, fn, flops, grid_hint=""):
    getter = getattr(lib, f"insar_debug_{which}_stamps")
    getter.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    getter(None, 1)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    getter(buf.ctypes.data, 0)
    s = buf.reshape(1024, 8).astype(np.float64).sum(0) / reps
    clock = ""
    if which in ("igemm", "wgrad3x") and s[7] > 0:          # [6] shader-clock ticks, [7] 100 MHz ticks of the same interval, summed over work-groups
        ghz = s[6] / s[7] * 0.1
        clock = f"clock {ghz:.2f} GHz -> dense bf16 peak {2500.0 * ghz / 2.4:.0f} TF, kernel at {100 * flops / us / 1e6 / (2500.0 * ghz / 2.4):.0f}% of it | "
        s = s.copy(); s[6] = s[7] = 0
    tot = s.sum()
    return (f"{us:7.1f} us {flops / us / 1e6:6.0f} TF {grid_hint}| " + clock +
            " | ".join(f"{names[k]} {100 * s[k] / tot:.1f}%" for k in range(8) if names[k] != "-"))


def main():
    lib = _lib.load()
    if not hasattr(lib, "insar_debug_igemm_stamps"):
        sys.exit("needs the diagnostic build: INSAR_HIP_LIB=.../libinsar_hip_stamps.so")
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    ctx.side = None
    for name, (cin, cout, hw) in LAYERS.items():
        B = 16
        x = engine.Act.alloc(B, hw, hw, cin, dtype, dev); x.buf[:, 1:-1, 1:-1].normal_()
        y = engine.Act.alloc(B, hw, hw, cout, dtype, dev)
        g = engine.Act.alloc(B, hw, hw, cout, dtype, dev); g.buf[:, 1:-1, 1:-1].normal_()
        dx = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
        p = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        gw = engine.GemmWeight(ctx, p, "conv3")
        wf, wd = gw.fwd(), gw.dgrad()
        M = B * hw * hw
        flops = 2.0 * M * cin * cout * 9
        stats = torch.zeros(call("insar_igemm_num_mtiles", M, cout), 2, cout, device=dev)
        tile = lambda n: f"{call('insar_igemm_tile_rows', M, n)}x{call('insar_igemm_tile_cols_dt', M, n, _lib.BF16)}"
        print(f"{name} {cin}->{cout} @{hw}^2")
        print("   fwd   ", report(lib, "igemm", IG, lambda: engine._igemm(x, y, wf, cout, hw, hw, 1, engine._TAPS3, 0, stats=stats), flops, f"tile {tile(cout)} "))
        print("   dgrad ", report(lib, "igemm", IG, lambda: engine._igemm(g, dx, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0), flops, f"tile {tile(cin)} "))
        pair = call("insar_wgrad_conv3_tile", x.ref, cout)
        if pair:
            tm, tn = pair >> 16, pair & 0xffff
            tiles = 3 * (cin // tm) * (cout // tn)
            for fill in (1.0, 0.5):
                nsp = engine._wgrad_nsplit(tiles, M // 64, 9 * cout * cin, tm, tn, 2, taps_per_wg=3, fill=fill)
                if fill < 1.0 and tm * tn >= 128 * 128:
                    nsp = min(nsp, max(1, engine.WGRAD_GRID_CAP // tiles))
                part = ctx.wgrad_part(nsp * 9 * cout * cin)
                print(f"   wgrad3 fill {fill}", report(lib, "wgrad3", W3, lambda: call("insar_wgrad_conv3", x.ref, g.ref, ptr(part), nsp, _lib.stream_ptr()), flops,
                                                      f"tile {tm}x{tn} grid {tiles * nsp} "), flush=True)
        pairx = call("insar_wgrad_conv3x_tile", x.ref, cout)
        if pairx:          # the 256 x 128 six-phase kernel (csrc/wgrad3x.hip)
            tm, tn = pairx >> 16, pairx & 0xffff
            tiles = 3 * (cin // tm) * (cout // tn)
            for fill in (1.0, 0.6):
                nsp = engine._wgrad_nsplit(tiles, M // 64, 9 * cout * cin, tm, tn, 2, taps_per_wg=3, fill=fill)
                if fill < 1.0:
                    nsp = min(nsp, max(1, engine.WGRAD_GRID_CAP // tiles))
                part = ctx.wgrad_part(nsp * 9 * cout * cin)
                print(f"   wgrad3x fill {fill}", report(lib, "wgrad3x", W3, lambda: call("insar_wgrad_conv3x", x.ref, g.ref, ptr(part), nsp, _lib.stream_ptr()), flops,
                                                       f"tile {tm}x{tn} grid {tiles * nsp} "), flush=True)


if __name__ == "__main__":
    main()
