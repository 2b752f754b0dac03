"""Where a tile of the flat 3x3 kernel spends its cycles: in-kernel s_memtime stamps (diagnostic build of the library,
csrc/conv3x3_flat.hip under -DINSAR_STAMPS -> insar_unet_ca_amd/libinsar_hip_stamps.so; the product library has no stamps).
usage: INSAR_HIP_LIB=insar_unet_ca_amd/libinsar_hip_stamps.so python tools/stamp_flat.py
Per layer shape and direction, persistent and one-tile-per-work-group launches of the 8-wave kernel, then the two-work-group
kernel (csrc/conv3x3_flat2.hip; persistent, flat geometry and row tiles: a work-group's stamps are its OWN serial timeline —
what the partner on the same CU hides is the difference between the sum of the shares and the launch time): cycles per tile of wave 0 in each phase,
averaged over the work-groups (s_memtime ticks at 100 MHz on gfx950: reported as a share of the tile and in ns)."""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from insar_unet_ca_amd import engine, _lib
from insar_unet_ca_amd._lib import call, ptr

PHASES = ["tile geometry", "issue first slabs", "row offsets", "first slabs landed", "K loop", "epilogue (own)", "epilogue (wait others)", "turn-around"]
LAYERS = {"down1.0": (64, 128, 128), "down1.3": (128, 128, 128), "conv4.0": (128, 64, 256)}


def main():
    lib = _lib.load()
    if not hasattr(lib, "insar_debug_flat_stamps"):
        sys.exit("needs the diagnostic build: INSAR_HIP_LIB=.../libinsar_hip_stamps.so")
    lib.insar_debug_flat_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.insar_debug_flat2_stamps.argtypes = [C.c_void_p, C.c_int]
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    for name, (cin, cout, hw) in LAYERS.items():
        B = 16
        x = engine.Act.alloc(B, hw, hw, cin, dtype, dev); x.buf[:, 1:-1, 1:-1].normal_()
        y = engine.Act.alloc(B, hw, hw, cout, dtype, dev)
        g = engine.Act.alloc(B, hw, hw, cout, dtype, dev); g.buf[:, 1:-1, 1:-1].normal_()
        dx = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
        p = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        gw = engine.GemmWeight(ctx, p, "conv3")
        wf, wd = gw.fwd(), gw.dgrad()
        for direction, (src, dst, w, flip, n) in {"fwd": (x, y, wf, 0, cout), "dgrad": (g, dx, wd, 1, cin)}.items():
            for flags, getter, label in ((2 | 4, lib.insar_debug_flat_stamps, "8-wave persistent"), (2, lib.insar_debug_flat_stamps, "8-wave per tile"),
                                         (32 | 4, lib.insar_debug_flat2_stamps, "2 x 4-wave persistent"), (32 | 8 | 4, lib.insar_debug_flat2_stamps, "2 x 4-wave row tiles persistent")):
                rows = call("insar_conv3x3_flat_stat_rows", src.ref, n, flags)
                st = torch.zeros(rows, 2, n, device=dev)
                mt = (B * hw * hw // 256) if flags & 8 else call("insar_conv3x3_flat_num_mtiles", src.ref)
                tiles = mt * (n // (128 if n % 128 == 0 else 64))
                fn = lambda: call("insar_conv3x3_flat", src.ref, dst.ref, ptr(w), flip | flags, ptr(st) if flip == 0 else 0, _lib.stream_ptr())
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
                slots = 512 if flags & 32 else 256
                grid = min(tiles, slots) if flags & 4 else tiles
                s = buf.reshape(1024, 8).astype(np.float64).sum(0) / reps      # ticks per launch, summed over the work-groups
                per_tile = s / tiles
                tot = per_tile.sum()
                print(f"{name} {cin}->{cout} @{hw}^2 {direction} {label}: {us:.1f} us/launch, {tiles} tiles, {tiles / grid:.1f} per work-group; "
                      f"wave-0 ticks per tile {tot:.0f} (x10 ns)")
                print("    " + " | ".join(f"{PHASES[k]} {100 * per_tile[k] / tot:.1f}% ({10 * per_tile[k]:.0f} ns)" for k in range(8)), flush=True)


if __name__ == "__main__":
    main()
