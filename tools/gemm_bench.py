"""Micro-benchmark of the GEMM-class kernels on U-Net-CA layer shapes (B=16, 256x256 tiles).
usage: python tools/gemm_bench.py [--iters N] [--only name,...] [--dtype bf16|f32]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from insar_unet_ca_amd import engine, _lib
from insar_unet_ca_amd._lib import call

LAYERS = {  # name: (cin, cout, hw)
    "inc.3": (64, 64, 256), "down1.0": (64, 128, 128), "down1.3": (128, 128, 128), "down2.3": (256, 256, 64),
    "down3.3": (512, 512, 32), "down4.3": (1024, 1024, 16), "conv1.0": (1024, 512, 32), "conv2.0": (512, 256, 64),
    "conv3.0": (256, 128, 128), "conv4.0": (128, 64, 256),
    "down2.0": (128, 256, 64), "down3.0": (256, 512, 32), "down4.0": (512, 1024, 16), "conv1.3": (512, 512, 32), "conv2.3": (256, 256, 64),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--what", default="fwd,dgrad,flat,wgrad")
    ap.add_argument("--knob", default="", help="with --what knob-wgrad / knob-fwd / knob-dgrad: name=v0,v1[,v2...] library knob to A/B")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    ctx = engine.Ctx(dev, dtype)
    ctx.side = None          # kernels measured alone: weight gradients get the split-K factor that fills the chip
    names = [n for n in a.only.split(",") if n] or list(LAYERS)
    what = a.what.split(",")
    for name in names:
        cin, cout, hw = LAYERS[name]
        B = a.B
        x = engine.Act.alloc(B, hw, hw, cin, dtype, dev); x.buf[:, 1:-1, 1:-1].normal_()
        y = engine.Act.alloc(B, hw, hw, cout, dtype, dev)
        g = engine.Act.alloc(B, hw, hw, cout, dtype, dev); g.buf[:, 1:-1, 1:-1].normal_()
        dx = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
        p = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        gw = engine.GemmWeight(ctx, p, "conv3")
        M = B * hw * hw
        stats = torch.zeros(call("insar_igemm_num_mtiles", M, cout), 2, cout, device=dev)
        grad = torch.zeros(cout, cin, 3, 3, device=dev)
        wf, wd = gw.fwd(), gw.dgrad()
        flops = 2.0 * M * cin * cout * 9
        def run(fn):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters * 1e3
        res = []
        if "pp" in what:       # same-process A/B of the ping-pong K loop (256 x 256 tiles only): interleaved rounds
            rounds = {0: [], 1: []}
            for r in range(4):
                for pp in (0, 1):
                    engine.IGEMM_PP = pp
                    rounds[pp].append((run(lambda: engine._igemm(x, y, wf, cout, hw, hw, 1, engine._TAPS3, 0, stats=stats)),
                                       run(lambda: engine._igemm(g, dx, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0))))
            engine.IGEMM_PP = 0
            for pp in (0, 1):
                f = sorted(v[0] for v in rounds[pp]); d = sorted(v[1] for v in rounds[pp])
                res.append(f"pp{pp}: fwd min {f[0]:6.1f} med {f[len(f)//2]:6.1f} us ({flops/f[0]/1e6:6.0f} TF) dgrad min {d[0]:6.1f} med {d[len(d)//2]:6.1f} us ({flops/d[0]/1e6:6.0f} TF)")
        if "fwd" in what:
            us = run(lambda: engine._igemm(x, y, wf, cout, hw, hw, 1, engine._TAPS3, 0, stats=stats)); res.append(f"fwd {us:7.1f} us {flops/us/1e6:7.1f} TF")
        if "dgrad" in what:
            us = run(lambda: engine._igemm(g, dx, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0)); res.append(f"dgrad {us:7.1f} us {flops/us/1e6:7.1f} TF")
        if "fpp" in what:      # same-process A/B of the flat kernel's ping-pong tap steps
            from insar_unet_ca_amd._lib import ptr
            rows = call("insar_conv3x3_flat_num_mtiles", x.ref)
            st2 = torch.zeros(rows, 2, cout, device=dev)
            rounds = {0: [], 2: []}
            for r in range(4):
                for pp in (0, 2):
                    rounds[pp].append((run(lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), 0 | pp, ptr(st2), _lib.stream_ptr())),
                                       run(lambda: call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), 1 | pp, 0, _lib.stream_ptr()))))
            outs = {}
            for pp in (0, 2):
                y.buf.zero_(); dx.buf.zero_(); st2.zero_()
                call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), 0 | pp, ptr(st2), _lib.stream_ptr())
                call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), 1 | pp, 0, _lib.stream_ptr()); torch.cuda.synchronize()
                outs[pp] = (y.buf.clone(), dx.buf.clone(), st2.clone())
                f = sorted(v[0] for v in rounds[pp]); d = sorted(v[1] for v in rounds[pp])
                res.append(f"flat pp{pp}: fwd min {f[0]:6.1f} med {f[len(f)//2]:6.1f} us ({flops/f[0]/1e6:6.0f} TF) dgrad min {d[0]:6.1f} med {d[len(d)//2]:6.1f} us ({flops/d[0]/1e6:6.0f} TF)")
            res.append("bitwise=" + str(all(torch.equal(a_, b_) for a_, b_ in zip(outs[0], outs[2]))))
        if "fpers" in what:    # persistent tile loop of the flat kernel (flip bit 4), same-process A/B, interleaved rounds
            from insar_unet_ca_amd._lib import ptr
            rows = call("insar_conv3x3_flat_num_mtiles", x.ref)
            st2 = torch.zeros(rows, 2, cout, device=dev)
            rr = {0: [], 4: []}
            for r in range(4):
                for pb in (0, 4):
                    rr[pb].append((run(lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), 0 | 2 | pb, ptr(st2), _lib.stream_ptr())),
                                   run(lambda: call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), 1 | 2 | pb, 0, _lib.stream_ptr()))))
            for pb in (0, 4):
                f = sorted(v[0] for v in rr[pb]); d = sorted(v[1] for v in rr[pb])
                res.append(f"\n   persist={pb >> 2}: fwd min {f[0]:6.1f} med {f[len(f)//2]:6.1f} us  dgrad min {d[0]:6.1f} med {d[len(d)//2]:6.1f} us")
        if "rows" in what:     # per-tap igemm against the flat kernel's row tiles (flip bit 3), same process, interleaved rounds
            from insar_unet_ca_amd._lib import ptr
            if call("insar_conv3x3_flat_rows_ok", x.ref, cout) and call("insar_conv3x3_flat_rows_ok", g.ref, cin):
                for narrow in (0, 16):
                    fl = 8 | 2 | narrow
                    st3 = torch.zeros(call("insar_conv3x3_flat_stat_rows", x.ref, cout, fl), 2, cout, device=dev)
                    rr = {"igemm": [], "rows": [], "flat": []}
                    flat_ok = call("insar_conv3x3_flat_ok", x.ref, cout) and call("insar_conv3x3_flat_ok", g.ref, cin) and not narrow
                    if flat_ok:
                        st4 = torch.zeros(call("insar_conv3x3_flat_stat_rows", x.ref, cout, 2 | 4), 2, cout, device=dev)
                    for r in range(4):
                        if flat_ok:      # the flat pixel-space geometry as the step runs it: ping-pong, persistent
                            rr["flat"].append((run(lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), 2 | 4, ptr(st4), _lib.stream_ptr())),
                                               run(lambda: call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), 2 | 4 | 1, 0, _lib.stream_ptr()))))
                        rr["igemm"].append((run(lambda: engine._igemm(x, y, wf, cout, hw, hw, 1, engine._TAPS3, 0, stats=stats)),
                                            run(lambda: engine._igemm(g, dx, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0))))
                        rr["rows"].append((run(lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), fl, ptr(st3), _lib.stream_ptr())),
                                           run(lambda: call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), fl | 1, 0, _lib.stream_ptr()))))
                    for k in ("igemm", "rows", "flat"):
                        if not rr[k]:
                            continue
                        f = sorted(v[0] for v in rr[k]); d = sorted(v[1] for v in rr[k])
                        res.append(f"\n   {k:5s} narrow={narrow >> 4}: fwd min {f[0]:6.1f} med {f[len(f)//2]:6.1f} us ({flops/f[0]/1e6:5.0f} TF)  dgrad min {d[0]:6.1f} med {d[len(d)//2]:6.1f} us ({flops/d[0]/1e6:5.0f} TF)")
        if "flat2" in what:      # the two-work-group build of the flat kernel against the 8-wave one, as the engine launches them
            from insar_unet_ca_amd._lib import ptr
            import ctypes as C
            rr = {}
            yb = engine.Act.alloc(B, hw, hw, cin, dtype, dev); yb.buf[:, 1:-1, 1:-1].normal_()
            sc, sh = torch.randn(cin, device=dev), torch.randn(cin, device=dev) * 0.3
            bs = _lib.InsarBstat(yb.buf.data_ptr(), ptr(sc), ptr(sh))
            for r in range(3):
                for nm, fl in (("8-wave persistent", 2 | 4), ("8-wave row tiles", 2 | 8), ("8-wave row tiles 64-col", 2 | 8 | 16), ("2 x 4-wave persistent", 32 | 4),
                               ("2 x 4-wave row tiles", 32 | 8 | 4), ("2 x 4-wave row tiles 64-col", 32 | 8 | 4 | 16)):
                    if (fl & 8) and not (call("insar_conv3x3_flat2_rows_ok", x.ref, cout) and call("insar_conv3x3_flat_rows_ok", x.ref, cout) and call("insar_conv3x3_flat_rows_ok", g.ref, cin)):
                        continue
                    rows = call("insar_conv3x3_flat_stat_rows", x.ref, cout, fl)
                    st = torch.zeros(rows, 2, cout, device=dev)
                    rowsb = call("insar_conv3x3_flat_stat_rows", g.ref, cin, fl)
                    stb = torch.zeros(rowsb, 2, cin, device=dev)
                    rr.setdefault(nm, []).append((
                        run(lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), fl, ptr(st), _lib.stream_ptr())),
                        run(lambda: call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), fl | 1, 0, _lib.stream_ptr())),
                        run(lambda: call("insar_conv3x3_flat_bstat", g.ref, dx.ref, ptr(wd), fl | 1, ptr(stb), C.byref(bs), _lib.stream_ptr()))))
            for nm, v in rr.items():
                best = [min(t[i] for t in v) for i in range(3)]
                res.append(f"\n   {nm:28s} fwd+stats {best[0]:6.1f} us ({flops/best[0]/1e6:5.0f} TF)  dgrad {best[1]:6.1f} us ({flops/best[1]/1e6:5.0f} TF)  dgrad+bstat {best[2]:6.1f} us ({flops/best[2]/1e6:5.0f} TF)")
        if "flat" in what:
            from insar_unet_ca_amd._lib import ptr
            rows = call("insar_conv3x3_flat_num_mtiles", x.ref)
            st2 = torch.zeros(rows, 2, cout, device=dev)
            us = run(lambda: call("insar_conv3x3_flat", x.ref, y.ref, ptr(wf), 0, ptr(st2), _lib.stream_ptr())); res.append(f"flat-fwd {us:7.1f} us {flops/us/1e6:7.1f} TF")
            us = run(lambda: call("insar_conv3x3_flat", g.ref, dx.ref, ptr(wd), 1, 0, _lib.stream_ptr())); res.append(f"flat-dgrad {us:7.1f} us {flops/us/1e6:7.1f} TF")
        for mode in [w for w in what if w.startswith("knob-")]:
            # same-process A/B of a library knob (insar_tune_set): interleaved rounds, min / median, results compared
            kname, _, vals = a.knob.partition("=")
            vals = [int(v) for v in vals.split(",")]
            fn = {"knob-wgrad": lambda: call("insar_wgrad_conv3", x.ref, g.ref, _lib.ptr(part_), nsp_, _lib.stream_ptr()),
                  "knob-fwd": lambda: engine._igemm(x, y, wf, cout, hw, hw, 1, engine._TAPS3, 0, stats=stats),
                  "knob-dgrad": lambda: engine._igemm(g, dx, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0)}[mode]
            if mode == "knob-wgrad":
                pair = call("insar_wgrad_conv3_tile", x.ref, cout)
                if not pair:
                    continue
                tm, tn = pair >> 16, pair & 0xffff
                tiles = 3 * (cin // tm) * (cout // tn)
                nsp_ = engine._wgrad_nsplit(tiles, M // 64, 9 * cout * cin, tm, tn, 2, taps_per_wg=3, fill=1.0)
                part_ = ctx.wgrad_part(nsp_ * 9 * cout * cin)
            rounds = {v: [] for v in vals}
            outs = {}
            for r in range(5):
                for v in vals:
                    _lib.tune(kname, v)
                    rounds[v].append(run(fn))
            for v in vals:
                _lib.tune(kname, v)
                if mode == "knob-wgrad":
                    part_.zero_(); fn(); torch.cuda.synchronize(); outs[v] = part_[:nsp_ * 9 * cout * cin].clone()
                elif mode == "knob-fwd":
                    y.buf.zero_(); fn(); torch.cuda.synchronize(); outs[v] = y.buf.float().clone()
                else:
                    dx.buf.zero_(); fn(); torch.cuda.synchronize(); outs[v] = dx.buf.float().clone()
                t = sorted(rounds[v])
                res.append(f"\n   {mode} {kname}={v}: min {t[0]:6.1f} med {t[len(t)//2]:6.1f} us ({flops/t[0]/1e6:6.0f} / {flops/t[len(t)//2]/1e6:6.0f} TF)")
            _lib.tune(kname, vals[0])
            ref = outs[vals[0]]
            for v in vals[1:]:
                d = (outs[v] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
                res.append(f"max|{kname}={v} - {kname}={vals[0]}|/max = {d:.2e}" + (" (bitwise)" if torch.equal(outs[v], ref) else ""))
        if "y3" in what:       # wgrad3y.hip (128 x 128 tiles, two 4-wave work-groups per CU) against wgrad3x.hip (256 x 128, one 8-wave group per CU)
            pairx = call("insar_wgrad_conv3x_tile", x.ref, cout)
            if not pairx and call("insar_wgrad_conv3y_tile", x.ref, cout) and call("insar_wgrad_conv3_tile", x.ref, cout):
                # 128 x 128 layers: against the 8-wave 128-tile kernel (wgrad3.hip)
                tiles = 3 * (cin // 128) * (cout // 128)
                ks = M // 64
                part_ = ctx.wgrad_part((512 // tiles + 1) * 9 * cout * cin)
                def f3(n): return lambda: call("insar_wgrad_conv3", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                def fy(n): return lambda: call("insar_wgrad_conv3y", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                part_.zero_(); f3(2)(); torch.cuda.synchronize(); ref = part_[:2 * 9 * cout * cin].clone()
                part_.zero_(); fy(2)(); torch.cuda.synchronize(); got = part_[:2 * 9 * cout * cin].clone()
                res.append(f"bitwise@nsplit=2: {torch.equal(ref, got)}")
                for cus in (256, 128):
                    n3 = max(1, min(cus // tiles, ks // 4)); ny = max(1, min(2 * cus // tiles, ks // 4))
                    r3, ry = [], []
                    for r in range(4):
                        r3.append(run(f3(n3))); ry.append(run(fy(ny)))
                    res.append(f"\n   {cus} CUs: wgrad3 128x128 nsplit {n3:3d} grid {n3 * tiles:4d}: {min(r3):6.1f} us ({flops/min(r3)/1e6:5.0f} TF) | "
                               f"wgrad3y 128x128 nsplit {ny:3d} grid {ny * tiles:4d}: {min(ry):6.1f} us ({flops/min(ry)/1e6:5.0f} TF)")
            if pairx and call("insar_wgrad_conv3y_tile", x.ref, cout):
                tmx, tnx = pairx >> 16, pairx & 0xffff
                tiles_x = 3 * (cin // tmx) * (cout // tnx)
                tiles_y = 3 * (cin // 128) * (cout // 128)
                ks = M // 64
                part_ = ctx.wgrad_part(max(1, 512 // tiles_y + 1, 256 // tiles_x + 1) * 9 * cout * cin)
                def fx(n): return lambda: call("insar_wgrad_conv3x", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                def fy(n): return lambda: call("insar_wgrad_conv3y", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                part_.zero_(); fx(1)(); torch.cuda.synchronize(); ref = part_[:9 * cout * cin].clone()
                part_.zero_(); fy(1)(); torch.cuda.synchronize(); got = part_[:9 * cout * cin].clone()
                res.append(f"bitwise@nsplit=1: {torch.equal(ref, got)}")
                for cus in (256, 128):          # the share of the chip the launch aims at: whole / half (as beside the dgrad chain)
                    nx = max(1, min(cus // tiles_x, ks // 4)); ny = max(1, min(2 * cus // tiles_y, ks // 4))
                    rx, ry = [], []
                    for r in range(4):
                        rx.append(run(fx(nx))); ry.append(run(fy(ny)))
                    res.append(f"\n   {cus} CUs: wgrad3x {tmx}x{tnx} nsplit {nx:3d} grid {nx * tiles_x:4d}: {min(rx):6.1f} us ({flops/min(rx)/1e6:5.0f} TF) | "
                               f"wgrad3y 128x128 nsplit {ny:3d} grid {ny * tiles_y:4d}: {min(ry):6.1f} us ({flops/min(ry)/1e6:5.0f} TF)")
        if "x3" in what:       # the 256 x 128 six-phase weight-gradient kernel (wgrad3x.hip) against the 128 x 128 row-of-taps one
            pairx = call("insar_wgrad_conv3x_tile", x.ref, cout)
            pair3 = call("insar_wgrad_conv3_tile", x.ref, cout)
            if pairx and pair3:
                tmx, tnx = pairx >> 16, pairx & 0xffff
                tm3, tn3 = pair3 >> 16, pair3 & 0xffff
                tiles_x = 3 * (cin // tmx) * (cout // tnx)
                tiles_3 = 3 * (cin // tm3) * (cout // tn3)
                ks = M // 64
                cands = []
                for fill in (1.0, 0.6):
                    nx = engine._wgrad_nsplit(tiles_x, ks, 9 * cout * cin, tmx, tnx, 2, taps_per_wg=3, fill=fill)
                    n3 = engine._wgrad_nsplit(tiles_3, ks, 9 * cout * cin, tm3, tn3, 2, taps_per_wg=3, fill=fill)
                    if fill < 1.0 and engine.WGRAD_GRID_CAP:
                        nx = min(nx, max(1, engine.WGRAD_GRID_CAP // tiles_x)); n3 = min(n3, max(1, engine.WGRAD_GRID_CAP // tiles_3))
                    cands.append((fill, n3, nx))
                extra = sorted({max(1, 128 // tiles_x), max(1, 192 // tiles_x), max(1, 256 // tiles_x), max(1, 512 // tiles_x)})
                part_ = ctx.wgrad_part(max(max(c[1], c[2]) for c in cands + [(0, 1, e) for e in extra]) * 9 * cout * cin)
                def fx(n): return lambda: call("insar_wgrad_conv3x", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                def f3(n): return lambda: call("insar_wgrad_conv3", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                def fin(n): return lambda: ctx.wgrad_finish(part_, grad, n, 9, cout, cin, 0)
                # bitwise at equal nsplit
                nchk = cands[0][2]
                part_.zero_(); f3(nchk)(); torch.cuda.synchronize(); ref = part_[:nchk * 9 * cout * cin].clone()
                part_.zero_(); fx(nchk)(); torch.cuda.synchronize(); got = part_[:nchk * 9 * cout * cin].clone()
                res.append(f"bitwise@nsplit={nchk}: {torch.equal(ref, got)} (max diff {(ref - got).abs().max().item():.2e})")
                for fill, n3, nx in cands:
                    r3, rx, q3, qx = [], [], [], []
                    for r in range(4):
                        r3.append(run(f3(n3))); rx.append(run(fx(nx))); q3.append(run(fin(n3))); qx.append(run(fin(nx)))
                    res.append(f"\n   fill {fill}: wgrad3 {tm3}x{tn3} nsplit {n3:3d} grid {n3 * tiles_3:4d}: {min(r3):6.1f} us ({flops/min(r3)/1e6:5.0f} TF) + fold {min(q3):5.1f} | "
                               f"wgrad3x {tmx}x{tnx} nsplit {nx:3d} grid {nx * tiles_x:4d}: {min(rx):6.1f} us ({flops/min(rx)/1e6:5.0f} TF) + fold {min(qx):5.1f}")
                for n in extra:
                    t = min(run(fx(n)) for _ in range(3)); q = min(run(fin(n)) for _ in range(2))
                    res.append(f"\n   wgrad3x nsplit {n:3d} grid {n * tiles_x:4d} steps/wg {ks / n:6.1f}: {t:6.1f} us ({flops/t/1e6:5.0f} TF) + fold {q:5.1f}")
        if "k3" in what:       # the pixel-slice weight-gradient kernel (wgrad3k.hip) against the row-of-taps kernel of wgrad3.hip
            pairk = call("insar_wgrad_conv3k_tile", x.ref, cout)
            pair3 = call("insar_wgrad_conv3_tile", x.ref, cout)
            if pairk and pair3:
                tmk, tnk = pairk >> 16, pairk & 0xffff
                ksl = call("insar_wgrad_conv3k_slices", x.ref, cout)
                tm3, tn3 = pair3 >> 16, pair3 & 0xffff
                tiles_k = 3 * (cin // tmk) * (cout // tnk)
                tiles_3 = 3 * (cin // tm3) * (cout // tn3)
                nmax = max(1, 512 // tiles_k)
                part_ = ctx.wgrad_part(max(nmax * ksl, 1024) * 9 * cout * cin)
                def fk(n): return lambda: call("insar_wgrad_conv3k", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                def f3(n): return lambda: call("insar_wgrad_conv3", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                def fin(n): return lambda: ctx.wgrad_finish(part_, grad, n, 9, cout, cin, 0)
                # against the old kernel's folded gradient
                n3 = engine._wgrad_nsplit(tiles_3, M // 64, 9 * cout * cin, tm3, tn3, 2, taps_per_wg=3, fill=1.0)
                f3(n3)(); fin(n3)(); torch.cuda.synchronize(); ref = grad.clone()
                nk = max(1, 256 // tiles_k)
                fk(nk)(); fin(nk * ksl)(); torch.cuda.synchronize(); got = grad.clone()
                res.append(f"max|k - 3|/max|3| = {(got - ref).abs().max().item() / ref.abs().max().item():.2e}")
                for fill in (1.0, 0.6):
                    n3 = engine._wgrad_nsplit(tiles_3, M // 64, 9 * cout * cin, tm3, tn3, 2, taps_per_wg=3, fill=fill)
                    nk = max(1, int(256 * fill) // tiles_k)
                    r3 = min(run(f3(n3)) for _ in range(3)); q3 = min(run(fin(n3)) for _ in range(2))
                    rk = min(run(fk(nk)) for _ in range(3)); qk = min(run(fin(nk * ksl)) for _ in range(2))
                    res.append(f"\n   fill {fill}: wgrad3 {tm3}x{tn3} nsplit {n3:3d} grid {n3 * tiles_3:4d}: {r3:6.1f} us ({flops/r3/1e6:5.0f} TF) + fold {q3:5.1f} | "
                               f"wgrad3k {tmk}x{tnk} (x{ksl} slices) nsplit {nk:3d} grid {nk * tiles_k:4d}: {rk:6.1f} us ({flops/rk/1e6:5.0f} TF) + fold {qk:5.1f}")
                for n in sorted({max(1, 128 // tiles_k), max(1, 192 // tiles_k), max(1, 384 // tiles_k), nmax}):
                    t = min(run(fk(n)) for _ in range(3))
                    res.append(f"\n   wgrad3k nsplit {n:3d} grid {n * tiles_k:4d}: {t:6.1f} us ({flops/t/1e6:5.0f} TF)")
        if "x3var" in what:    # timing ablations of wgrad3x's K loop (experiment build: make exp EXPNAME=wx EXPFLAGS=-DINSAR_EXP_WX; INSAR_HIP_LIB=...)
            pairx = call("insar_wgrad_conv3x_tile", x.ref, cout)
            if pairx and (pairx >> 16) == 256:
                tiles_x = 3 * (cin // 256) * (cout // 128)
                n = max(1, 256 // tiles_x)
                part_ = ctx.wgrad_part(n * 9 * cout * cin)
                fn = lambda: call("insar_wgrad_conv3x", x.ref, g.ref, _lib.ptr(part_), n, _lib.stream_ptr())
                names_ = {0: "kernel", 1: "no DMA in loop", 4: "lockstep groups", 8: "no MFMA",
                          9: "no MFMA, no DMA", 17: "no reads, no DMA (MFMA + barriers)", 25: "barriers only"}
                rr = {v: [] for v in names_}
                for r in range(3):
                    for v in names_:
                        _lib.tune("wgrad3x_var", v); rr[v].append(run(fn))
                _lib.tune("wgrad3x_var", 0)
                steps = (M // 64) / n
                for v in names_:
                    t = min(rr[v])
                    res.append(f"\n   var {v:2d} {names_[v]:36s}: {t:6.1f} us  ({t / steps * 1e3:6.0f} ns per K step, grid {n * tiles_x}, {steps:.1f} steps/wg)")
        if "wgrad" in what:
            us = run(lambda: engine._wgrad_conv3(ctx, x, g, grad)); res.append(f"wgrad(+fold) {us:7.1f} us {flops/us/1e6:7.1f} TF")
        print(f"{name:9s} {cin:4d}->{cout:4d} @{hw:3d}^2 tile_rows {call('insar_igemm_tile_rows', M, cout)}: " + " | ".join(res), flush=True)

if __name__ == "__main__":
    main()
