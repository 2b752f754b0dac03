"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the drop-in module
surface and the C ABI, against (a) the golden vectors generated from the reference itself
(tests/golden, see oracle/gen_golden.py), (b) the CPU oracle on the same seeded inputs, and (c)
size-independent properties at the full BASELINE configuration.

Tolerances
  forward, fp32 : max-rel <= 1e-3 (north_star); observed ~1e-5.
  backward, fp32: two fp32 implementations disagree on the ReLU mask of any pre-activation that sits
                  within rounding of 0, which changes that element's gradient by O(1). So:
                  kernel-level gradient tests (no ReLU) are tight (<= 2e-5 max-rel against float64 on
                  the kernel's own operands); block/network gradient tests use rel-L2 with a floor
                  derived from torch's own fp32-vs-fp64 disagreement on the same fixture.
  bf16          : calibrated against torch's own bf16 on the same fixture (max-rel 0.08-0.10,
                  98.3-98.7 % argmax agreement, measured in-container); the HIP path measures 8.2e-2 / 98.5 % on G3 and
                  5.8e-2 / 98.6 % on G3r: gates 0.12 / 98 % (G3) and 8.5e-2 / 98 % (G3r).
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import closed_form as cf
from oracle import unet_ca_oracle as orc
from tests.helpers import check_grad_summary, check_summary, max_rel, to_np

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-3          # north_star
KERNEL_TOL = 2e-5
BF16_FWD_TOL = 0.12          # G3 fixtures: measured 8.2e-2 (torch's own bf16 on them: 0.08-0.10); 1.5x measured
BF16_ARGMAX = 0.98           # measured 0.9854 (torch's own: 0.983-0.987)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from insar_unet_ca_amd import _lib
    _lib.load()                      # fail loudly if the extension is missing
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = to_np(a), to_np(b)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


def _filled(mod):
    mod.load_state_dict(cf.fill_state_dict(mod.state_dict()))
    return mod


def _act_from(x, dtype, dev):
    from insar_unet_ca_amd import engine
    b, c, h, w = x.shape
    a = engine.Act.alloc(b, h, w, c, dtype, dev)
    engine.pack_input(x.to(dev), a)
    return a


def _halo_abs(a):
    t = a.buf.float().clone()
    t[:, 1:-1, 1:-1] = 0
    return float(t.abs().max())


# ------------------------------------------------------------------------------------------------
# kernel level: implicit GEMM forward / dgrad / wgrad, transposed conv — tight, no ReLU involved
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,shape", [(128, 128, (4, 128, 128, 128)), (64, 128, (8, 64, 100, 90)), (128, 64, (2, 128, 256, 256))])
def test_flat_kernel_pingpong_is_bitwise_the_plain_loop(dev, cin, cout, shape):
    """conv3x3_flat's ping-pong tap steps (flip bit 1), alone and with persistent work-groups (bit 2: one per CU walking
    several tiles through the same LDS), against its plain one-tile-per-work-group loop: forward with BatchNorm partial sums
    and input gradient, bit for bit, 20 times over each."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input_random(shape, seed=5), dtype, dev)
    ga = _act_from(cf.make_input_random((b, cout, h, w), seed=6), dtype, dev)
    assert call("insar_conv3x3_flat_ok", xa.ref, cout) == 1
    gw = engine.GemmWeight(ctx, torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev)), "conv3")
    wf, wd = gw.fwd(), gw.dgrad()
    rows = call("insar_conv3x3_flat_num_mtiles", xa.ref)
    ref, persistent_stats = None, None
    for pp, reps in ((0, 1), (2, 20), (4, 20), (6, 20)):
        for _ in range(reps):
            ya = engine.Act.alloc(b, h, w, cout, dtype, dev)
            dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
            stats = torch.zeros(rows, 2, cout, device=dev)
            call("insar_conv3x3_flat", xa.ref, ya.ref, ptr(wf), 0 | pp, ptr(stats), _lib.stream_ptr())
            call("insar_conv3x3_flat", ga.ref, dxa.ref, ptr(wd), 1 | pp, 0, _lib.stream_ptr())
            if ref is None:
                ref = (ya.buf.clone(), dxa.buf.clone(), stats.clone())
                assert float(ref[0].float().abs().max()) > 0
            elif pp & 4:
                # persistent work-groups carry the sums over their tiles: one slab row per work-group, the rest untouched
                assert torch.equal(ya.buf, ref[0]) and torch.equal(dxa.buf, ref[1])
                prow = call("insar_conv3x3_flat_stat_rows", xa.ref, cout, pp)
                assert prow <= rows and float(stats[prow:].abs().max() if prow < rows else 0.0) == 0.0
                assert max_rel(stats.sum(0), ref[2].sum(0)) <= 1e-5
                if persistent_stats is None:
                    persistent_stats = {}
                assert torch.equal(persistent_stats.setdefault(pp, stats.clone()), stats)      # run-to-run: bit for bit
            else:
                assert torch.equal(ya.buf, ref[0]) and torch.equal(dxa.buf, ref[1]) and torch.equal(stats, ref[2])


@pytest.mark.parametrize("cin,cout,shape", [(64, 512, (2, 64, 128, 128)), (256, 64, (4, 256, 128, 128)), (256, 256, (16, 256, 64, 64)),
                                            (1024, 256, (16, 1024, 32, 32))])
def test_pingpong_k_loop_is_bitwise_the_plain_loop(dev, cin, cout, shape, monkeypatch):
    """The 256 x 256-tile kernel's ping-pong K loop (INSAR_IGEMM_PINGPONG) accumulates every output in the same order as the
    plain two-slab loop: forward (with BatchNorm partial sums) and input gradient must agree bit for bit, many times over
    (its LDS hand-off between the two wave groups is the kind of code whose races come and go)."""
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input_random(shape, seed=3), dtype, dev)
    ga = _act_from(cf.make_input_random((b, cout, h, w), seed=4), dtype, dev)
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    rows = call("insar_igemm_num_mtiles", b * h * w, cout)
    assert 256 in (call("insar_igemm_tile_cols_dt", b * h * w, cout, 1), call("insar_igemm_tile_cols_dt", b * h * w, cin, 1))
    outs = {}
    for pp in (0, 1):
        monkeypatch.setattr(engine, "IGEMM_PP", pp)
        ya = engine.Act.alloc(b, h, w, cout, dtype, dev)
        dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
        stats = torch.zeros(rows, 2, cout, device=dev)
        for _ in range(25 if pp else 1):
            engine._igemm(xa, ya, gw.fwd(), cout, h, w, 1, engine._TAPS3, 0, stats=stats)
            engine._igemm(ga, dxa, gw.dgrad(), cin, h, w, 1, engine._TAPS3_DGRAD, 0)
            if pp:
                assert torch.equal(ya.buf, outs[0][0]) and torch.equal(dxa.buf, outs[0][1]) and torch.equal(stats, outs[0][2])
        outs[pp] = (ya.buf.clone(), dxa.buf.clone(), stats.clone())
    assert float(outs[1][0].float().abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,shape", [
    (64, 128, (2, 64, 16, 16)),
    (128, 64, (1, 128, 8, 24)),      # 192 pixels: partial M tile
    (256, 256, (2, 256, 8, 8)),
    (64, 64, (3, 64, 4, 4)),         # 48 pixels: M smaller than one tile
    (64, 64, (3, 64, 120, 200)),     # 72000 pixels: 256-row tiles (3-slab ring), ragged last tile
    (64, 128, (4, 64, 128, 128)),    # 65536 pixels: 256-row tiles, N = 128
    (256, 512, (4, 256, 32, 32)),    # 256x256 weight-gradient tiles (8 waves) with split-K, two Cout tiles
    (64, 1024, (4, 64, 32, 32)),     # 4096 pixels x 1024 channels: the 256x64 tile choice
    (256, 128, (2, 256, 16, 24)),    # weight gradient: Cin allows 256, Cout only 128 -> 128 x 128 tiles
    (128, 256, (3, 128, 16, 16)),
    (64, 64, (2, 64, 32, 64)),       # W % 64 == 0: row-of-taps weight gradient, 64 x 64 tile (bf16)
    (128, 128, (2, 128, 24, 128)),   # row-of-taps weight gradient, 128 x 128 tile with 8 waves, two K steps per row
    (64, 512, (2, 64, 128, 128)),    # forward N = 512 on 128 M tiles: 256 x 256 tiles (bf16)
    (256, 64, (4, 256, 128, 128)),   # input gradient N = 256 on 256 M tiles: 256 x 256 tiles (bf16)
    (128, 64, (2, 128, 64, 64)),     # row-of-taps weight gradient, 128 x 64 tile
    (64, 64, (2, 64, 32, 32)),       # row-of-taps weight gradient with two image rows per K step
])
def test_igemm_family_against_float64(dev, dtype, cin, cout, shape):
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input(shape), dtype, dev)
    ya = engine.Act.alloc(b, h, w, cout, dtype, dev)
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    stats = torch.zeros(call("insar_igemm_num_mtiles", b * h * w, cout), 2, cout, device=dev)
    engine._igemm(xa, ya, gw.fwd(), cout, h, w, 1, engine._TAPS3, 0, stats=stats)
    xr = xa.nchw().cpu().double()                         # operands exactly as the kernel saw them
    wr = gw.fwd().float().cpu().reshape(3, 3, cout, cin).permute(2, 3, 0, 1).double()
    ref = F.conv2d(xr, wr, padding=1)
    tol = KERNEL_TOL if dtype == torch.float32 else 6e-3   # bf16: only the output rounding remains
    assert max_rel(ya.nchw(), ref) <= tol
    assert _halo_abs(ya) == 0.0
    got = ya.nchw().double().cpu()
    assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
    assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4
    ga = _act_from(cf.make_grad((b, cout, h, w)), dtype, dev)
    dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
    engine._igemm(ga, dxa, gw.dgrad(), cin, h, w, 1, engine._TAPS3_DGRAD, 0)
    gr = ga.nchw().cpu().double()
    assert max_rel(dxa.nchw(), F.conv_transpose2d(gr, wr, padding=1)) <= tol
    gwt = torch.zeros(cout, cin, 3, 3, device=dev)
    engine._wgrad_conv3(ctx, xa, ga, gwt)
    wv = wr.clone().requires_grad_(True)
    F.conv2d(xr, wv, padding=1).backward(gr)
    assert max_rel(gwt, wv.grad) <= KERNEL_TOL * 5          # fp32 accumulation in both dtypes


@pytest.mark.parametrize("cin,cout,shape,c_extra", [
    (256, 128, (2, 256, 8, 64), 0),      # W = 64: a K step is one image row; 16 K steps
    (256, 128, (1, 256, 4, 128), 64),    # two K steps per image row; x is a channel slice of a wider buffer
    (512, 256, (2, 512, 32, 32), 0),     # two image rows per K step (W = 32), 2 x 2 tiles
    (256, 256, (4, 256, 16, 16), 0),     # four image rows per K step (W = 16)
    (128, 256, (2, 128, 16, 64), 0),     # 128 x 256 tiles (only Cout has 256)
    (256, 128, (1, 256, 1, 64), 0),      # ONE K step in all
    (256, 128, (1, 256, 2, 64), 0),      # two K steps
])
def test_wgrad_conv3x_against_float64_and_the_128_tile_kernel(dev, cin, cout, shape, c_extra):
    """csrc/wgrad3x.hip (256 x 128 tiles, six-phase K loop, three-slot LDS ring with a counted wait): the weight gradient
    against float64 on the kernel's own operands, and its split-K slabs BIT FOR BIT those of insar_wgrad_conv3 at the same
    nsplit — for split factors that give one, two, three and many K steps per work-group, ragged last splits and splits
    with no step at all (the ring's prologue / steady state / drain paths)."""
    from insar_unet_ca_amd import engine, _lib
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xfull = _act_from(cf.make_input((b, cin + c_extra, h, w)), dtype, dev)
    xa = xfull.slice(c_extra, cin) if c_extra else xfull
    ga = _act_from(cf.make_grad((b, cout, h, w)), dtype, dev)
    pair = call("insar_wgrad_conv3x_tile", xa.ref, cout)
    assert pair and call("insar_wgrad_conv3_tile", xa.ref, cout)
    ksteps = b * h * w // 64
    xr = xa.nchw().cpu().double()
    gr = ga.nchw().cpu().double()
    wv = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wv, padding=1).backward(gr)
    slab = 9 * cout * cin
    for nsplit in sorted({1, 2, 3, max(1, ksteps // 3), max(1, ksteps // 2), ksteps, ksteps + 3, max(1, (ksteps + 4) // 5)}):
        pa = torch.full((nsplit * slab,), float("nan"), device=dev)
        pb = torch.full((nsplit * slab,), float("nan"), device=dev)
        call("insar_wgrad_conv3", xa.ref, ga.ref, ptr(pa), nsplit, _lib.stream_ptr())
        call("insar_wgrad_conv3x", xa.ref, ga.ref, ptr(pb), nsplit, _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(pa, pb), (nsplit, float((pa - pb).abs().max()))
        # csrc/wgrad3y.hip: 128 x 128 tiles, 4-wave work-groups (two per CU), two-slot ring, one barrier per K step
        assert call("insar_wgrad_conv3y_tile", xa.ref, cout) == (128 << 16) | 128
        pc = torch.full((nsplit * slab,), float("nan"), device=dev)
        call("insar_wgrad_conv3y", xa.ref, ga.ref, ptr(pc), nsplit, _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(pa, pc), ("wgrad3y", nsplit, float((pa - pc).abs().max()))
        gwt = torch.zeros(cout, cin, 3, 3, device=dev)
        ctx.wgrad_finish(pb, gwt, nsplit, 9, cout, cin, 0)
        assert max_rel(gwt, wv.grad) <= KERNEL_TOL * 5, nsplit
    # and through the engine's own dispatch (cost-model split factor)
    gwt = torch.zeros(cout, cin, 3, 3, device=dev)
    engine._wgrad_conv3(ctx, xa, ga, gwt)
    assert max_rel(gwt, wv.grad) <= KERNEL_TOL * 5


@pytest.mark.parametrize("cin,cout,shape,nsplit", [
    (512, 256, (16, 512, 64, 64), 12),   # conv2.0 of config 2 in the step's launch configuration (85 K steps per work-group)
    (1024, 1024, (16, 1024, 16, 16), 2), # down4.3: four image rows per K step, 32 steps per work-group
    (256, 128, (4, 256, 32, 64), 9),     # ragged split: 128 K steps over 9 splits
])
@pytest.mark.parametrize("entry", ["insar_wgrad_conv3x", "insar_wgrad_conv3y"])
def test_wgrad_conv3x_race_screen(dev, cin, cout, shape, nsplit, entry):
    """The six-phase K loop is a NEW synchronisation structure (three-slot ring, one counted vmcnt per step, two wave groups a
    barrier apart): a read that runs ahead of the wait that retires its slot passes every reference check whenever the DMA
    happens to land first. Screen: the same launch 150 times, half of them beside a stream that keeps the memory system busy
    (the situation in which round 1's WAR race showed), every slab bit for bit the first one. Also for csrc/wgrad3y.hip (two-slot
    ring, one barrier per step, a second work-group of the same launch on the CU)."""
    from insar_unet_ca_amd import engine, _lib
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    b, _, h, w = shape
    xa = engine.Act.alloc(b, h, w, cin, dtype, dev)
    ga = engine.Act.alloc(b, h, w, cout, dtype, dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    xa.buf[:, 1:-1, 1:-1] = torch.randn((b, h, w, cin), device=dev, generator=gen).to(dtype)
    ga.buf[:, 1:-1, 1:-1] = torch.randn((b, h, w, cout), device=dev, generator=gen).to(dtype)
    assert call("insar_wgrad_conv3x_tile", xa.ref, cout)
    n = nsplit * 9 * cout * cin
    ref = torch.empty(n, device=dev)
    call(entry, xa.ref, ga.ref, ptr(ref), nsplit, _lib.stream_ptr())
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    out = torch.empty(n, device=dev)
    bad = 0
    for it in range(150):
        if it % 2:
            with torch.cuda.stream(side):
                noise.copy_(noise.flip(0))                        # 128 MB of traffic beside the launch
        out.fill_(float("nan"))
        call(entry, xa.ref, ga.ref, ptr(out), nsplit, _lib.stream_ptr())
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    assert bad == 0, f"{bad} of 150 launches differ from the first"


@pytest.mark.parametrize("cin,cout,shape,c_extra", [
    (64, 64, (1, 64, 3, 256), 0),        # 64 x 64: eight pixel slices, a K step = one 256-pixel image row; two ring slots
    (64, 64, (2, 64, 2, 512), 64),       # two K steps per image row; x is a channel slice of a wider buffer
    (128, 64, (2, 128, 4, 128), 0),      # 128 x 64: four slices of two wave tiles, three ring slots
    (64, 128, (1, 64, 5, 256), 0),       # 64 x 128
    (128, 128, (2, 128, 6, 64), 0),      # 128 x 128: two slices of four wave tiles
    (128, 128, (1, 128, 1, 64), 0),      # ONE K step in all
])
def test_wgrad_conv3k_against_float64(dev, cin, cout, shape, c_extra):
    """csrc/wgrad3k.hip (64 / 128 channels a side: the waves of a work-group split the pixels of a K step, a work-group writes
    KS slabs): the folded weight gradient against float64 on the kernel's own operands, for split factors that give one, two,
    three and many K steps per work-group, ragged last splits and splits without a step; bit for bit reproducible."""
    from insar_unet_ca_amd import engine, _lib
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xfull = _act_from(cf.make_input((b, cin + c_extra, h, w)), dtype, dev)
    xa = xfull.slice(c_extra, cin) if c_extra else xfull
    ga = _act_from(cf.make_grad((b, cout, h, w)), dtype, dev)
    pair = call("insar_wgrad_conv3k_tile", xa.ref, cout)
    ks = call("insar_wgrad_conv3k_slices", xa.ref, cout)
    assert pair == ((128 if cin % 128 == 0 else 64) << 16 | (128 if cout % 128 == 0 else 64)) and ks == 8 // ((pair >> 16) // 64 * ((pair & 0xffff) // 64))
    ksteps = b * h * w // (ks * 32)
    xr = xa.nchw().cpu().double()
    gr = ga.nchw().cpu().double()
    wv = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wv, padding=1).backward(gr)
    slab = 9 * cout * cin
    for nsplit in sorted({1, 2, 3, max(1, ksteps // 3), max(1, ksteps // 2), ksteps, ksteps + 2}):
        part = torch.full((nsplit * ks * slab,), float("nan"), device=dev)
        call("insar_wgrad_conv3k", xa.ref, ga.ref, ptr(part), nsplit, _lib.stream_ptr())
        again = torch.full((nsplit * ks * slab,), float("nan"), device=dev)
        call("insar_wgrad_conv3k", xa.ref, ga.ref, ptr(again), nsplit, _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(part, again), nsplit
        gwt = torch.zeros(cout, cin, 3, 3, device=dev)
        ctx.wgrad_finish(part, gwt, nsplit * ks, 9, cout, cin, 0)
        assert max_rel(gwt, wv.grad) <= KERNEL_TOL * 5, nsplit
    gwt = torch.zeros(cout, cin, 3, 3, device=dev)
    engine._wgrad_conv3(ctx, xa, ga, gwt)                       # the engine's own dispatch
    assert max_rel(gwt, wv.grad) <= KERNEL_TOL * 5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,shape", [
    (64, 128, (2, 64, 40, 56)),      # several 254-pixel tiles, rows shorter than a tile
    (128, 64, (1, 128, 33, 300)),    # rows longer than a tile (a tile inside one image row), N = 64
    (64, 64, (3, 64, 30, 30)),       # tiles straddle image boundaries
    (256, 128, (1, 256, 32, 32)),    # four K slabs per tap
])
def test_conv3x3_flat_against_float64(dev, dtype, cin, cout, shape):
    """The flat-padded 3x3 kernel (forward and flipped = input gradient), called directly so that small
    grids exercise it too; the plans only select it when the grid fills the chip."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input(shape), dtype, dev)
    ya = engine.Act.alloc(b, h, w, cout, dtype, dev)
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    rows = call("insar_conv3x3_flat_num_mtiles", xa.ref)
    stats = torch.zeros(rows, 2, cout, device=dev)
    call("insar_conv3x3_flat", xa.ref, ya.ref, ptr(gw.fwd()), 0, ptr(stats), _lib.stream_ptr())
    xr = xa.nchw().cpu().double()
    wr = gw.fwd().float().cpu().reshape(3, 3, cout, cin).permute(2, 3, 0, 1).double()
    tol = KERNEL_TOL if dtype == torch.float32 else 6e-3
    assert max_rel(ya.nchw(), F.conv2d(xr, wr, padding=1)) <= tol
    assert _halo_abs(ya) == 0.0
    got = ya.nchw().double().cpu()
    assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
    assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4
    ga = _act_from(cf.make_grad((b, cout, h, w)), dtype, dev)
    dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
    call("insar_conv3x3_flat", ga.ref, dxa.ref, ptr(gw.dgrad()), 1, 0, _lib.stream_ptr())
    assert max_rel(dxa.nchw(), F.conv_transpose2d(ga.nchw().cpu().double(), wr, padding=1)) <= tol
    assert _halo_abs(dxa) == 0.0


@pytest.mark.parametrize("cin,cout,shape", [
    (64, 128, (2, 64, 40, 56)),      # several 254-pixel tiles, rows shorter than a tile; two 32-channel slabs
    (128, 64, (1, 128, 33, 300)),    # rows longer than a tile, 64-column tiles (64 x 64 wave tiles)
    (32, 64, (3, 32, 30, 30)),       # one slab: the loop is its tail only; tiles straddle image boundaries
    (128, 128, (4, 128, 128, 128)),  # the 128^2 level's shape: more tiles than persistent work-groups (carried sums)
    (256, 128, (1, 256, 32, 32)),    # eight slabs per tap
    (64, 128, (3, 64, 4, 128)),      # row tiles, W = 128: two image rows per tile, every wave's pixels one image row
    (128, 64, (2, 128, 3, 256)),     # row tiles, W = 256, 64-column tiles: one image row per tile, four waves along it
    (64, 64, (10, 64, 128, 128)),    # row tiles: 640 tiles for 512 persistent work-groups (carried sums, uneven shares)
    (128, 128, (3, 128, 8, 64)),     # row tiles, W = 64: four image rows per tile, two per wave (two edge fragments a side)
    (64, 128, (2, 64, 16, 32)),      # row tiles, W = 32: eight image rows per tile
    (64, 64, (3, 64, 16, 16)),       # row tiles, W = 16: every fragment begins and ends an image row; 64-column tiles
])
def test_conv3x3_flat_two_work_group_kernel_against_float64(dev, cin, cout, shape):
    """csrc/conv3x3_flat2.hip (flip bit 5: two co-resident 4-wave work-groups per CU, 32-channel slabs, 128 x 64 wave tiles),
    one tile per work-group and persistent (bit 2), in the flat geometry and — where the grid is 128 / 256 pixels wide — as row
    tiles (bit 3: the halo pixels are not staged, the edge lanes' fragments are cleared): forward with BatchNorm partial sums and
    input gradient against float64,
    zero halos, against the 8-wave kernel to bf16 rounding of its fp32 sums (the summation order differs), and bit for bit
    run to run over 10 launches each (the second work-group of a CU runs beside the first: a race screen too)."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input_random(shape, seed=5), dtype, dev)
    ga = _act_from(cf.make_input_random((b, cout, h, w), seed=6), dtype, dev)
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    wr = gw.fwd().float().cpu().reshape(3, 3, cout, cin).permute(2, 3, 0, 1).double()
    want_y = F.conv2d(xa.nchw().cpu().double(), wr, padding=1)
    want_dx = F.conv_transpose2d(ga.nchw().cpu().double(), wr, padding=1)
    rows = call("insar_conv3x3_flat_num_mtiles", xa.ref)
    y8 = engine.Act.alloc(b, h, w, cout, dtype, dev)
    if cin % 64 == 0:
        call("insar_conv3x3_flat", xa.ref, y8.ref, ptr(gw.fwd()), 2, 0, _lib.stream_ptr())
    rows_ok = call("insar_conv3x3_flat2_rows_ok", xa.ref, cout)
    assert rows_ok == (1 if (w in (16, 32, 64, 128, 256) and h % (256 // w) == 0) else 0)
    for flags in (32, 32 | 4) + ((32 | 8, 32 | 8 | 4) if rows_ok else ()):
        first = None
        rows = b * h * w // 256 if flags & 8 else call("insar_conv3x3_flat_num_mtiles", xa.ref)
        for rep in range(10):
            ya = engine.Act.alloc(b, h, w, cout, dtype, dev)
            dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
            prow = call("insar_conv3x3_flat_stat_rows", xa.ref, cout, flags)
            assert prow <= rows
            stats = torch.zeros(rows, 2, cout, device=dev)
            call("insar_conv3x3_flat", xa.ref, ya.ref, ptr(gw.fwd()), flags, ptr(stats), _lib.stream_ptr())
            if cin % 64 == 0:          # (the input gradient has cin output columns: multiples of 64)
                call("insar_conv3x3_flat", ga.ref, dxa.ref, ptr(gw.dgrad()), flags | 1, 0, _lib.stream_ptr())
            if first is None:
                first = (ya.buf.clone(), dxa.buf.clone(), stats.clone())
                assert max_rel(ya.nchw(), want_y) <= 6e-3, flags
                assert cin % 64 or max_rel(dxa.nchw(), want_dx) <= 6e-3, flags
                assert _halo_abs(ya) == 0.0 and _halo_abs(dxa) == 0.0
                got = ya.nchw().double().cpu()
                assert float(stats[prow:].abs().max() if prow < rows else 0.0) == 0.0
                assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
                assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4
                if cin % 64 == 0:
                    assert max_rel(ya.nchw(), y8.nchw().double().cpu()) <= 8e-3      # one bf16 ulp of the largest output
            else:
                assert torch.equal(ya.buf, first[0]) and torch.equal(dxa.buf, first[1]) and torch.equal(stats, first[2]), (flags, rep)


def test_conv3x3_flat_two_work_group_kernel_beyond_2_gib(dev):
    """An activation buffer larger than 2^31 bytes (64 x 256 x 256 tiles of 256 channels: 2.18 GB — a batch the 288 GB of one
    MI355X invite): the two-work-group kernel addresses its operands by a 64-bit tile base + 32-bit lane offsets relative to
    the tile, in both geometries; checked against the 8-wave kernel (64-bit lane addresses) to one bf16 rounding of the largest
    output, on the last image too — the one beyond the 2 GiB mark."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, cin, cout, h, w = 64, 256, 64, 256, 256
    xa = engine.Act.alloc(b, h, w, cin, dtype, dev)
    assert xa.buf.numel() * 2 > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(3)
    for i in range(b):
        xa.buf[i, 1:-1, 1:-1] = torch.randn(h, w, cin, device=dev, generator=g).to(dtype)
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    wf = engine.GemmWeight(ctx, p, "conv3").fwd()
    ref = engine.Act.alloc(b, h, w, cout, dtype, dev)
    call("insar_conv3x3_flat", xa.ref, ref.ref, ptr(wf), 2 | 8, 0, _lib.stream_ptr())
    scale = float(ref.buf.float().abs().max())
    assert scale > 0
    for flags in (32 | 4, 32 | 8 | 4, 32 | 8):
        ya = engine.Act.alloc(b, h, w, cout, dtype, dev)
        call("insar_conv3x3_flat", xa.ref, ya.ref, ptr(wf), flags, 0, _lib.stream_ptr())
        for img in (0, b // 2, b - 1):
            d = float((ya.buf[img].float() - ref.buf[img].float()).abs().max())
            assert d <= 8e-3 * scale, (flags, img, d, scale)
        assert _halo_abs(ya) == 0.0
        del ya


@pytest.mark.parametrize("cin,cout,shape,dil,narrow", [
    (64, 128, (2, 64, 32, 32), 2, False),      # DeepLabV3 layer3 / layer4 geometry: 8 rows per tile, 36 staged columns
    (128, 64, (2, 128, 32, 32), 4, True),      # dilation 4: 40 staged columns, 320 staged rows (the A slot exactly), 64-column tiles
    (64, 64, (1, 64, 16, 64), 2, False),       # 4 rows per tile, dilation 2
    (64, 128, (3, 64, 8, 32), 3, False),       # an odd dilation, one tile per image
])
def test_conv3x3_flat_dilated_row_tiles_against_float64(dev, cin, cout, shape, dil, narrow):
    """Row tiles of a dilated 3x3 convolution (padding = dilation; flip bits 8-11): taps that reach beyond the one-pixel halo
    of the activation buffer read zeros. Forward with BatchNorm partial sums and input gradient against float64 and against
    the per-tap kernel's out-of-bounds variant (same operands; different accumulation order: output rounding apart)."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input(shape), dtype, dev)
    assert call("insar_conv3x3_flat_rows_dil_ok", xa.ref, cout, dil) == 1
    assert call("insar_conv3x3_flat_rows_dil_ok", xa.ref, cout, 15) == 0            # 256 / W * (W + 30) > 320
    ya, yb = (engine.Act.alloc(b, h, w, cout, dtype, dev) for _ in range(2))
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    flags = 8 | 2 | (16 if narrow else 0) | (dil << 8)
    stats = torch.zeros(call("insar_conv3x3_flat_stat_rows", xa.ref, cout, flags), 2, cout, device=dev)
    call("insar_conv3x3_flat", xa.ref, ya.ref, ptr(gw.fwd()), flags, ptr(stats), _lib.stream_ptr())
    xr = xa.nchw().cpu().double()
    wr = gw.fwd().float().cpu().reshape(3, 3, cout, cin).permute(2, 3, 0, 1).double()
    ref = F.conv2d(xr, wr, padding=dil, dilation=dil)
    assert max_rel(ya.nchw(), ref) <= 6e-3
    assert _halo_abs(ya) == 0.0
    got = ya.nchw().double().cpu()
    assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
    assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4
    taps = [(ky * dil - dil, kx * dil - dil) for ky in range(3) for kx in range(3)]
    engine._igemm(xa, yb, gw.fwd(), cout, h, w, 1, taps, 0, oob=True)
    assert max_rel(ya.nchw(), yb.nchw().double().cpu()) <= 1.2e-2
    ga = _act_from(cf.make_grad((b, cout, h, w)), dtype, dev)
    dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
    call("insar_conv3x3_flat", ga.ref, dxa.ref, ptr(gw.dgrad()), flags | 1, 0, _lib.stream_ptr())
    assert max_rel(dxa.nchw(), F.conv_transpose2d(ga.nchw().cpu().double(), wr, padding=dil, dilation=dil)) <= 6e-3
    assert _halo_abs(dxa) == 0.0


@pytest.mark.parametrize("cin,cout,shape,narrow", [
    (128, 128, (2, 128, 32, 32), False),     # 8 image rows per tile, one N tile
    (256, 64, (4, 256, 16, 16), False),      # a tile = one whole 16 x 16 image, four K slabs per tap
    (64, 256, (1, 64, 64, 64), False),       # 4 rows per tile, two N tiles of 128
    (128, 256, (2, 128, 16, 16), True),      # 64-column tiles although N is a multiple of 128 (flag bit 4)
    (64, 64, (1, 64, 4, 128), False),        # two rows per tile
    (64, 128, (2, 64, 3, 256), False),       # one row per tile
])
def test_conv3x3_flat_row_tiles_against_float64(dev, cin, cout, shape, narrow):
    """The flat kernel's row-tile geometry (flip bit 3: 256 real pixels = whole image rows per tile, staged with their
    halo pixels; the three dx taps share the staged tile): forward with BatchNorm partial sums and input gradient
    against float64 on the kernel's own operands, bit for bit the flat geometry's output."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input(shape), dtype, dev)
    assert call("insar_conv3x3_flat_rows_ok", xa.ref, cout) == 1
    ya, yb = (engine.Act.alloc(b, h, w, cout, dtype, dev) for _ in range(2))
    p = torch.nn.Parameter(cf.fill_tensor("weight", (cout, cin, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    flags = 8 | 2 | (16 if narrow else 0)
    rows = call("insar_conv3x3_flat_stat_rows", xa.ref, cout, flags)
    assert rows == b * h * w // 256
    stats = torch.zeros(rows, 2, cout, device=dev)
    call("insar_conv3x3_flat", xa.ref, ya.ref, ptr(gw.fwd()), flags, ptr(stats), _lib.stream_ptr())
    xr = xa.nchw().cpu().double()
    wr = gw.fwd().float().cpu().reshape(3, 3, cout, cin).permute(2, 3, 0, 1).double()
    assert max_rel(ya.nchw(), F.conv2d(xr, wr, padding=1)) <= 6e-3
    assert _halo_abs(ya) == 0.0
    got = ya.nchw().double().cpu()
    assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
    assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4
    # same products in the same order as the flat geometry: identical output
    call("insar_conv3x3_flat", xa.ref, yb.ref, ptr(gw.fwd()), 2, 0, _lib.stream_ptr())
    assert torch.equal(ya.buf, yb.buf)
    ga = _act_from(cf.make_grad((b, cout, h, w)), dtype, dev)
    dxa = engine.Act.alloc(b, h, w, cin, dtype, dev)
    call("insar_conv3x3_flat", ga.ref, dxa.ref, ptr(gw.dgrad()), flags | 1, 0, _lib.stream_ptr())
    assert max_rel(dxa.nchw(), F.conv_transpose2d(ga.nchw().cpu().double(), wr, padding=1)) <= 6e-3
    assert _halo_abs(dxa) == 0.0


@pytest.mark.parametrize("shape", [
    (2, 64, 40, 56),       # several tiles, short rows
    (1, 64, 33, 300),      # rows longer than a tile
    (3, 64, 30, 30),       # tiles straddle images
    (1, 64, 16, 16),       # fewer pixels than two tiles
    (2, 64, 256, 256),     # the full-resolution level: one work-group walks many tiles, ring wraps
])
def test_conv3x3_c64_against_float64(dev, shape):
    """The persistent 64 -> 64 channel kernel (weights in registers, rolling LDS window), forward with
    BatchNorm partial sums and flipped = input gradient, against float64 on the kernel's own operands."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    b, _, h, w = shape
    xa = _act_from(cf.make_input(shape), dtype, dev)
    assert call("insar_conv3x3_c64_ok", xa.ref, 64) == 1
    ya = engine.Act.alloc(b, h, w, 64, dtype, dev)
    p = torch.nn.Parameter(cf.fill_tensor("weight", (64, 64, 3, 3), 11).to(dev))
    gw = engine.GemmWeight(ctx, p, "conv3")
    rows = call("insar_conv3x3_c64_rows", xa.ref)
    stats = torch.full((rows, 2, 64), float("nan"), device=dev)
    call("insar_conv3x3_c64", xa.ref, ya.ref, ptr(gw.fwd()), 0, ptr(stats), _lib.stream_ptr())
    xr = xa.nchw().cpu().double()
    wr = gw.fwd().float().cpu().reshape(3, 3, 64, 64).permute(2, 3, 0, 1).double()
    assert max_rel(ya.nchw(), F.conv2d(xr, wr, padding=1)) <= 6e-3      # bf16 output rounding only
    assert _halo_abs(ya) == 0.0
    got = ya.nchw().double().cpu()
    assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
    assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4
    ga = _act_from(cf.make_grad((b, 64, h, w)), dtype, dev)
    dxa = engine.Act.alloc(b, h, w, 64, dtype, dev)
    call("insar_conv3x3_c64", ga.ref, dxa.ref, ptr(gw.dgrad()), 1, 0, _lib.stream_ptr())
    assert max_rel(dxa.nchw(), F.conv_transpose2d(ga.nchw().cpu().double(), wr, padding=1)) <= 6e-3
    assert _halo_abs(dxa) == 0.0
    # a channel slice of a wider buffer as input and as output (the concat buffers)
    wide = engine.Act.alloc(b, h, w, 128, dtype, dev)
    engine.pack_input(cf.make_input(shape).to(dev), wide.slice(64, 64))
    out = engine.Act.alloc(b, h, w, 128, dtype, dev)
    call("insar_conv3x3_c64", wide.slice(64, 64).ref, out.slice(0, 64).ref, ptr(gw.fwd()), 0, 0, _lib.stream_ptr())
    assert torch.equal(out.slice(0, 64).nchw(), ya.nchw())
    assert float(out.slice(64, 64).nchw().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,shape", [(1, (2, 1, 16, 16)), (2, (3, 2, 24, 40)), (3, (1, 3, 16, 20)), (4, (2, 4, 16, 32))])
def test_first_layer_conv_against_float64(dev, dtype, cin, shape):
    """The direct first-layer conv (Cin <= 4; bf16 with Cin <= 3 runs on the matrix cores,
    rows not a multiple of 16 pixels included) and its statistics slab, against float64 on its own operands."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    b, _, h, w = shape
    xa = _act_from(cf.make_input(shape, 0.3), dtype, dev)
    ya = engine.Act.alloc(b, h, w, 64, dtype, dev)
    wt = cf.fill_tensor("weight", (64, cin, 3, 3), 5).to(dev)
    rows = call("insar_conv3x3_small_fwd_rows", xa.ref, ya.ref)
    stats = torch.full((rows, 2, 64), float("nan"), device=dev)
    call("insar_conv3x3_small_fwd", xa.ref, ptr(wt), ya.ref, ptr(stats), _lib.stream_ptr())
    wref = wt.double().cpu()          # the matrix-core version carries the fp32 weight as bf16 head + remainder
    ref = F.conv2d(xa.nchw().cpu().double(), wref, padding=1)
    assert max_rel(ya.nchw(), ref) <= (KERNEL_TOL if dtype == torch.float32 else 6e-3)
    assert _halo_abs(ya) == 0.0
    got = ya.nchw().double().cpu()
    assert max_rel(stats.sum(0)[0], got.sum((0, 2, 3))) <= 1e-4
    assert max_rel(stats.sum(0)[1], (got ** 2).sum((0, 2, 3))) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,shape", [
    (2, (2, 2, 24, 128)),    # bf16: the matrix-core kernel (W % 64 == 0), two K steps per image row
    (2, (1, 2, 5, 64)),      # fewer K steps than waves: most waves fold an empty partial sum
    (2, (3, 2, 16, 192)),    # three K steps per row, three images
    (2, (3, 2, 24, 40)),     # W % 64 != 0: the VALU kernel (packed FMAs, prefetched items)
    (1, (2, 1, 16, 64)),     # one input channel: VALU kernel in both dtypes
    (4, (2, 4, 16, 32)),
])
def test_first_layer_weight_gradient_against_float64(dev, dtype, cin, shape):
    """dW of the first conv (Cin <= 4): per-block partial rows folded by insar_colsum, against float64 on the operands
    exactly as the kernel saw them."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    b, _, h, w = shape
    ctx = engine.Ctx(dev, dtype)
    xa = _act_from(cf.make_input(shape, 0.3), dtype, dev)
    ga = _act_from(cf.make_grad((b, 64, h, w)), dtype, dev)
    nb = call("insar_conv3x3_small_wgrad_blocks", b, h)
    cols = 64 * cin * 9
    part = torch.full((nb, cols), float("nan"), device=dev)
    call("insar_conv3x3_small_wgrad", xa.ref, ga.ref, ptr(part), _lib.stream_ptr())
    gw = torch.zeros(64, cin, 3, 3, device=dev)
    ctx.colsum(part, gw, 1, nb, cols)
    xr = xa.nchw().cpu().double()
    wv = torch.zeros(64, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wv, padding=1).backward(ga.nchw().cpu().double())
    assert max_rel(gw, wv.grad) <= KERNEL_TOL * 5            # fp32 accumulation in both dtypes


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weight_relayout_single_and_batched(dev, dtype):
    """GEMM-operand forms of Conv2d / ConvTranspose2d weights: the per-weight kernel and the one-launch
    paired kernel (both forms from one read) give exactly the permutations of the fp32 master."""
    from insar_unet_ca_amd import engine
    ctx = engine.Ctx(dev, dtype)
    shapes = [("conv3", (128, 64, 3, 3)), ("conv3", (64, 192, 3, 3)), ("convT", (128, 64, 2, 2)), ("convT", (64, 64, 2, 2)),
              ("conv3", (96, 40, 3, 3)), ("conv3", (2, 256, 1, 1)), ("conv3", (33, 7, 3, 3)), ("conv3", (256, 128, 1, 1))]     # ragged 64 x 64 tiles, odd extents
    params = [torch.nn.Parameter(cf.fill_tensor("weight", shp, 3 + i).to(dev)) for i, (_, shp) in enumerate(shapes)]
    single = [engine.GemmWeight(ctx, p, kind) for p, (kind, _) in zip(params, shapes)]
    batched = [engine.GemmWeight(ctx, p, kind) for p, (kind, _) in zip(params, shapes)]
    engine.WeightSet(ctx, batched).refresh()
    for (kind, shp), p, a, b in zip(shapes, params, single, batched):
        w = p.detach().to(dtype)
        if kind == "conv3":     # (Co,Ci,3,3) -> [tap][co][ci] and [tap][ci][co]
            fwd = w.permute(2, 3, 0, 1).reshape(shp[2] * shp[3], shp[0], shp[1])
            dgr = w.permute(2, 3, 1, 0).reshape(shp[2] * shp[3], shp[1], shp[0])
        else:                   # (Ci,Co,2,2) -> [tap][co][ci] and [tap][ci][co]
            fwd = w.permute(2, 3, 1, 0).reshape(4, shp[1], shp[0])
            dgr = w.permute(2, 3, 0, 1).reshape(4, shp[0], shp[1])
        for got in (a.fwd(), b._buf["fwd"]):
            assert torch.equal(got, fwd), (kind, shp)
        for got in (a.dgrad(), b._buf["dgrad"]):
            assert torch.equal(got, dgr), (kind, shp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_against_golden_and_float64(dev, dtype, golden):
    from insar_unet_ca_amd import engine
    g2 = golden("g2_resample")
    ctx = engine.Ctx(dev, dtype)
    mod = _filled(torch.nn.ConvTranspose2d(128, 64, 2, 2)).to(dev)
    xa = _act_from(cf.make_input((2, 128, 8, 8)), dtype, dev)
    cat = engine.Act.alloc(2, 16, 16, 128, dtype, dev)
    up = engine.UpPlan(ctx, mod, xa, cat.slice(64, 64), "up")
    up.forward()
    out = cat.slice(64, 64).nchw()
    assert float(cat.slice(0, 64).nchw().abs().max()) == 0.0          # skip half of the concat untouched
    dcat = engine.Act.alloc(2, 16, 16, 128, dtype, dev)
    engine.pack_input(cf.make_grad((2, 64, 16, 16)).to(dev), dcat.slice(64, 64))
    sink = engine.GradSink(ctx, up.params())
    dx = engine.Act.alloc(2, 8, 8, 128, dtype, dev)
    up.backward(dcat.slice(64, 64), sink, dx)
    tol = FWD_TOL if dtype == torch.float32 else 4e-2
    check_summary(g2, "convT_128_64/step0/out", out, tol)
    check_summary(g2, "convT_128_64/step0/dx", dx.nchw(), tol)
    check_summary(g2, "convT_128_64/step0/grad/weight", sink.view(mod.weight), tol)
    check_summary(g2, "convT_128_64/step0/grad/bias", sink.view(mod.bias), tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_apply_with_fused_maxpool_is_bitwise_the_two_pass_result(dev, dtype):
    """insar_bn_relu_apply_pool == insar_bn_relu_apply followed by insar_maxpool2_fwd, bit for bit (rounding to
    the storage type is monotone, so max-then-round equals the pool of the rounded values)."""
    from insar_unet_ca_amd import _lib, engine
    from insar_unet_ca_amd._lib import call, ptr
    b, c, h, w = 3, 128, 12, 20
    ya = _act_from(cf.make_input((b, c, h, w), 0.8), dtype, dev)
    scale = cf.fill_tensor("weight", (c,), 2).to(dev).contiguous()
    shift = cf.fill_tensor("bias", (c,), 3).to(dev).contiguous()
    gate = torch.sigmoid(cf.make_grad((b, c))).to(dev).contiguous()
    s = _lib.stream_ptr()
    two = engine.Act.alloc(b, h, w, c, dtype, dev)
    two_p = engine.Act.alloc(b, h // 2, w // 2, c, dtype, dev)
    call("insar_bn_relu_apply", ya.ref, ptr(scale), ptr(shift), ptr(gate), two.ref, 1, s)
    call("insar_maxpool2_fwd", two.ref, two_p.ref, s)
    one = engine.Act.alloc(b, h, w, 2 * c, dtype, dev)             # written as a channel slice of a wider buffer
    one_p = engine.Act.alloc(b, h // 2, w // 2, c, dtype, dev)
    call("insar_bn_relu_apply_pool", ya.ref, ptr(scale), ptr(shift), ptr(gate), one.slice(0, c).ref, one_p.ref, 1, s)
    assert torch.equal(one.slice(0, c).nchw(), two.nchw()) and torch.equal(one_p.nchw(), two_p.nchw())
    assert float(one.slice(c, c).nchw().abs().max()) == 0.0 and _halo_abs(one_p) == 0.0


def test_maxpool_golden_with_ties(dev, golden):
    import insar_unet_ca_amd as iu
    g2 = golden("g2_resample")
    x = torch.from_numpy(g2["pool/x"]).to(dev).requires_grad_(True)
    out = iu.MaxPool2d(2)(x)
    out.backward(cf.make_grad(out.shape).to(dev))
    check_summary(g2, "pool/out", out, 0.0)
    check_summary(g2, "pool/dx", x.grad, 0.0)            # ties route to the first max, bit-exact


def test_outc_golden(dev, golden):
    from insar_unet_ca_amd import engine
    g2 = golden("g2_resample")
    ctx = engine.Ctx(dev, torch.float32)
    mod = _filled(torch.nn.Conv2d(64, 2, 1)).to(dev)
    xa = _act_from(cf.make_input((2, 64, 16, 16)), torch.float32, dev)
    oc = engine.OutConvPlan(ctx, mod, xa)
    logits = oc.forward()
    sink = engine.GradSink(ctx, oc.params())
    dx = engine.Act.alloc(2, 16, 16, 64, torch.float32, dev)
    oc.backward(cf.make_grad(logits.shape).to(dev), sink, dx)
    check_summary(g2, "outc_64_2/step0/out", logits, 1e-5)
    check_summary(g2, "outc_64_2/step0/dx", dx.nchw(), 1e-5)
    check_summary(g2, "outc_64_2/step0/grad/weight", sink.view(mod.weight), 1e-5)
    check_summary(g2, "outc_64_2/step0/grad/bias", sink.view(mod.bias), 1e-5)


# ------------------------------------------------------------------------------------------------
# block level against the golden vectors of the reference's own classes (G1)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,c,shape,salt", [("se64", 64, (2, 64, 8, 8), 0.0), ("se128", 128, (3, 128, 4, 4), 0.3)])
def test_se_layer_golden(dev, golden, tag, c, shape, salt):
    import insar_unet_ca_amd as iu
    g1 = golden("g1_blocks")
    mod = _filled(iu.SELayer(c)).to(dev)
    x = cf.make_input(shape, salt).to(dev).requires_grad_(True)
    out = mod(x)
    out.backward(cf.make_grad(out.shape).to(dev))
    check_summary(g1, f"{tag}/step0/out", out, 1e-5)
    check_summary(g1, f"{tag}/step0/dx", x.grad, 1e-5)
    check_summary(g1, f"{tag}/step0/grad/fc.0.weight", mod.fc[0].weight.grad, 1e-4)
    check_summary(g1, f"{tag}/step0/grad/fc.2.weight", mod.fc[2].weight.grad, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag,c,shape", [("cam256", 256, (2, 256, 8, 8)), ("cam64_ties", 64, (3, 64, 12, 20))])
def test_channel_attention_module_golden(dev, golden, tag, c, shape, dtype):
    """G7 (config 5's ChannelAttentionModule, vectors from the reference's class): forward, input gradient
    (mean branch spread over the image, max branch routed to the FIRST maximum in scan order, tie cases
    included) and both MLP weight gradients."""
    import insar_unet_ca_amd as iu
    g7 = golden("g7_cam")
    mod = _filled(iu.ChannelAttentionModule(c, 16)).to(dev)
    mod.compute_dtype = dtype
    assert list(mod.state_dict().keys()) == ["mlp.0.weight", "mlp.2.weight"]
    x0 = torch.from_numpy(g7[f"{tag}/x"]) if f"{tag}/x" in g7.files else cf.make_input(shape, 0.4)
    x = x0.to(dev).requires_grad_(True)
    out = mod(x)
    out.backward(cf.make_grad(out.shape).to(dev))
    if dtype == torch.float32:
        check_summary(g7, f"{tag}/step0/out", out, 1e-5)
        check_summary(g7, f"{tag}/step0/dx", x.grad, 1e-5)
        check_summary(g7, f"{tag}/step0/grad/mlp.0.weight", mod.mlp[0].weight.grad, 1e-4)
        check_summary(g7, f"{tag}/step0/grad/mlp.2.weight", mod.mlp[2].weight.grad, 1e-4)
    else:                                   # bf16 storage of x, out, dout, dx; fp32 pooling / MLP
        check_summary(g7, f"{tag}/step0/out", out, 1e-2)
        if tag == "cam256":
            check_summary(g7, f"{tag}/step0/dx", x.grad, 2e-2)
        else:   # rounding x to bf16 merges near-maxima into ties: the max-branch gradient may land on another pixel
            nrm = float(g7[f"{tag}/step0/dx/norm"])
            assert abs(float(x.grad.double().norm()) - nrm) <= 0.05 * nrm
        check_summary(g7, f"{tag}/step0/grad/mlp.0.weight", mod.mlp[0].weight.grad, 6e-2)
        check_summary(g7, f"{tag}/step0/grad/mlp.2.weight", mod.mlp[2].weight.grad, 6e-2)


def _relu_mask_disagreements(runner, oracle_masks):
    """Count pre-activations whose ReLU decision differs from the oracle's (values within rounding of 0)."""
    n = 0
    for u, ref in zip((runner.plan.u1, runner.plan.u2), oracle_masks):
        z = u.y.nchw() * u.scale.view(1, -1, 1, 1) + u.shift.view(1, -1, 1, 1)
        n += int(((z > 0).cpu() != ref).sum())
    return n


@pytest.mark.parametrize("tag,cin,cout,use_se,shape,salt,training,steps", [
    ("dc_2_64_se_train", 2, 64, True, (2, 2, 16, 16), 0.0, True, 2),
    ("dc_64_128_se_train", 64, 128, True, (2, 64, 16, 16), 0.0, True, 2),
    ("dc_64_128_se_eval", 64, 128, True, (2, 64, 16, 16), 0.0, False, 1),
    ("dc_128_64_plain_train", 128, 64, False, (2, 128, 16, 16), 0.0, True, 1),
    ("dc_128_64_se_ragged", 128, 64, True, (1, 128, 8, 24), 0.7, True, 1),
])
def test_double_conv_golden(dev, golden, tag, cin, cout, use_se, shape, salt, training, steps):
    import insar_unet_ca_amd as iu
    g1 = golden("g1_blocks")
    mod = _filled(iu.DoubleConv(cin, cout, use_se=use_se)).to(dev)
    mod.train(training)
    need_dx = cin > 4
    for s in range(steps):
        sd = OrderedDict(("blk." + k, v.detach().cpu().clone()) for k, v in mod.state_dict().items())
        for p in mod.parameters():
            p.grad = None
        x = cf.make_input(shape, salt).to(dev).requires_grad_(need_dx)
        out = mod(x)
        out.backward(cf.make_grad(out.shape).to(dev))
        pre = f"{tag}/step{s}"
        check_summary(g1, f"{pre}/out", out, FWD_TOL)
        assert max_rel(out, g1[f"{pre}/out/full"]) <= 1e-4 if f"{pre}/out/full" in g1.files else True
        for k, b in mod.named_buffers():
            if not k.endswith("num_batches_tracked"):
                check_summary(g1, f"{pre}/buf/{k}", b, FWD_TOL)
            else:
                assert int(b) == (s + 1 if training else 0)
        # ReLU decisions of the oracle (recomputed here) vs ours decide how tight the gradient gate is
        with torch.no_grad():
            y1 = F.conv2d(cf.make_input(shape, salt), sd["blk.double_conv.0.weight"], sd["blk.double_conv.0.bias"], padding=1)
            z1 = F.batch_norm(y1, sd["blk.double_conv.1.running_mean"].clone(), sd["blk.double_conv.1.running_var"].clone(),
                              sd["blk.double_conv.1.weight"], sd["blk.double_conv.1.bias"], training=training, eps=1e-5)
            y2 = F.conv2d(torch.relu(z1), sd["blk.double_conv.3.weight"], sd["blk.double_conv.3.bias"], padding=1)
            z2 = F.batch_norm(y2, sd["blk.double_conv.4.running_mean"].clone(), sd["blk.double_conv.4.running_var"].clone(),
                              sd["blk.double_conv.4.weight"], sd["blk.double_conv.4.bias"], training=training, eps=1e-5)
        runner = list(mod._plans.plans.values())[0][0]
        flips = _relu_mask_disagreements(runner, (z1 > 0, z2 > 0))
        gtol = 2e-4 if flips == 0 else 5e-2
        if need_dx:
            check_summary(g1, f"{pre}/dx", x.grad, gtol)
        for k, p in mod.named_parameters():
            if k in ("double_conv.0.bias", "double_conv.3.bias") and training:
                # exactly 0 on the HIP path (the batch mean cancels a pre-BN bias); the reference holds
                # summation noise here
                assert float(p.grad.abs().max()) == 0.0
                assert float(g1[f"{pre}/grad/{k}/absmax"]) < 1e-4
                continue
            check_summary(g1, f"{pre}/grad/{k}", p.grad, gtol)


# ------------------------------------------------------------------------------------------------
# whole network against the golden vectors of the reference UNet (G3), fp32
# ------------------------------------------------------------------------------------------------
def _unet(dev, use_se=True, cin=2, dtype=None):
    import insar_unet_ca_amd as iu
    net = _filled(iu.UNet(cin, 2, use_se=use_se, compute_dtype=dtype))
    return net.to(dev)


def _metrics_from_device(logits, tgt, dev):
    from insar_unet_ca_amd import _lib
    from insar_unet_ca_amd._lib import call, ptr
    b, k = logits.shape[0], logits.shape[1]
    hw = logits.shape[2] * logits.shape[3]
    counts = torch.zeros(3, k, dtype=torch.int64, device=dev)
    lg = logits.detach().contiguous()
    call("insar_confusion", ptr(lg), ptr(tgt), b, k, hw, 255, ptr(counts), _lib.stream_ptr())
    c = counts.cpu().numpy()
    return orc.metrics_from_counts(c[0], c[1], c[2])


# (conv / BN / convT tensors, SE fc tensors): gate on the relative deviation of a gradient tensor's NORM from the reference's on
# G3. The ragged fixture meets the 5 % of the G3r test; the two smooth closed-form ones do not, and not because of this path:
# torch's own fp32 and fp64 backward disagree by 0.07-0.15 rel-L2 on them (hundreds of pre-activations within rounding of 0,
# batch statistics over 8-16 values at the bottleneck). Measured here (round 4): b2_64_train worst 0.115 (down4 BN weight),
# SE fc 0.079; b1_256_train every conv / BN / convT tensor < 0.05, SE fc 0.218 (down4's first Linear: a sum of cancelling
# terms). Gates = 2 x measured: a layer that is WRONG is off by O(1).
G3_NORM_TOL = {"b2_64_train": (0.23, 0.16), "b3_48x80_train": (5e-2, 5e-2), "b1_256_train": (5e-2, 0.44)}


@pytest.mark.parametrize("tag,shape,training", [
    ("b2_64_train", (2, 2, 64, 64), True),
    ("b3_48x80_train", (3, 2, 48, 80), True),
    ("b1_256_eval", (1, 2, 256, 256), False),
    ("b1_256_train", (1, 2, 256, 256), True),
])
def test_unet_golden_fp32(dev, golden, tag, shape, training):
    import insar_unet_ca_amd as iu
    g3 = golden("g3_unet")
    net = _unet(dev)
    net.train(training)
    x = cf.make_input(shape).to(dev)
    tgt = cf.make_target((shape[0], shape[2], shape[3]), ignore_every=13).to(dev)
    crit = iu.CrossEntropyLoss(ignore_index=255)
    with torch.set_grad_enabled(training):
        logits = net(x)
        loss = crit(logits, tgt)
    check_summary(g3, f"{tag}/logits", logits, FWD_TOL)
    assert abs(float(loss.detach()) - float(g3[f"{tag}/loss"])) <= 1e-4 * max(1.0, abs(float(g3[f"{tag}/loss"])))
    m = _metrics_from_device(logits, tgt, dev)
    np.testing.assert_allclose([m[k] for k in ("acc", "miou", "mpa", "mf1")], g3[f"{tag}/metrics"], atol=2e-3)
    if not training:
        return
    loss.backward()
    for k, b in net.named_buffers():
        if not k.endswith("num_batches_tracked"):
            check_summary(g3, f"{tag}/buf/{k}", b, FWD_TOL)
    # Gradients on these smooth closed-form fixtures are ill-conditioned (torch's own fp32 and fp64 disagree
    # by 0.07-0.15 rel-L2: hundreds of pre-activations within rounding of 0, amplified by small-batch BN at
    # the bottleneck), so only the outermost layer is compared here; element-level gradient parity on these
    # very inputs is test_unet_fp32_gradients_given_equal_relu_decisions.
    for k, p in net.named_parameters():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            assert float(p.grad.abs().max()) == 0.0
    for k in ("outc.weight", "outc.bias"):
        nrm = float(g3[f"{tag}/grad/{k}/norm"])
        assert abs(float(dict(net.named_parameters())[k].grad.double().norm()) - nrm) <= 1e-3 * nrm, k
    # ... and the NORM of every gradient tensor against the reference-generated vector (the gate of the G3r test), so that a
    # wrong layer cannot hide behind the decision-conditioned test's modified oracle: a layer that is wrong is off by O(1),
    # the ill-conditioning of these fixtures moves norms by per cent. G3_NORM_TOL[tag] holds the gate (measured x 2 where
    # the 5 % of G3r is not reached; see the table's comment).
    worst = []
    for k, p in net.named_parameters():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            continue
        nrm = float(g3[f"{tag}/grad/{k}/norm"])
        dev_ = abs(float(p.grad.double().norm()) - nrm) / max(nrm, 1e-30)
        tol = G3_NORM_TOL[tag][1] if ".fc." in k else G3_NORM_TOL[tag][0]
        if dev_ > tol:
            worst.append((round(dev_, 4), k))
    assert not worst, sorted(worst, reverse=True)[:12]


def test_unet_fp32_gradients_golden_generic_position(dev, golden):
    """G3r (vectors generated from the reference's own classes on PCG64 weights and inputs): logits, loss
    and BN buffers match to 1e-4, every gradient NORM to 5 %, and a 5-step Adam run reproduces the
    reference's loss curve to 1e-3. Element-level gradient parity is the subject of
    test_unet_fp32_gradients_given_equal_relu_decisions: this fixture has 28 pre-activations with
    |z| < 1e-5, the HIP path decides 5 of them the other way than the reference (forward noise 6e-6 vs
    torch's 2.5e-6), and each such flip moves upstream gradients by ~2e-3 (SE fc gradients up to 2e-2)."""
    import insar_unet_ca_amd as iu
    g = golden("g3r_unet_random")
    net = iu.UNet(2, 2, True)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
    net = net.to(dev).train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    x = cf.make_input_random((2, 2, 64, 64), seed=11).to(dev)
    tgt = cf.make_target_random((2, 64, 64), seed=13, ignore_frac=0.05).to(dev)
    logits = net(x)
    loss = crit(logits, tgt)
    loss.backward()
    check_summary(g, "b2_64_train/logits", logits, 1e-4)
    assert abs(float(loss.detach()) - float(g["b2_64_train/loss"])) <= 1e-5
    for k, p in net.named_parameters():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            assert float(p.grad.abs().max()) == 0.0
            continue
        nrm = float(g[f"b2_64_train/grad/{k}/norm"])
        assert abs(float(p.grad.double().norm()) - nrm) <= 5e-2 * nrm, k
        # element level against the 64 stored samples of every gradient tensor (oracle/gen_golden.py): at most one
        # sampled element off by more than 1 % of the tensor's largest entry (SE fc: 5 %, sums of cancelling terms) —
        # the isolated-flip rule of tests/helpers.check_grad_summary
        check_grad_summary(g, f"b2_64_train/grad/{k}", p.grad, 5e-2 if ".fc." in k else 1e-2, what=k)
    for k, b in net.named_buffers():
        if not k.endswith("num_batches_tracked"):
            check_summary(g, f"b2_64_train/buf/{k}", b, 1e-4)
    # five Adam steps from the same start
    net = iu.UNet(2, 2, True)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
    net = net.to(dev).train()
    opt = iu.Adam(net.parameters(), lr=1e-4)
    losses = []
    for step in range(5):
        xs = cf.make_input_random((2, 2, 64, 64), seed=100 + step).to(dev)
        ts = cf.make_target_random((2, 64, 64), seed=200 + step).to(dev)
        opt.zero_grad()
        l = crit(net(xs), ts)
        l.backward()
        opt.step()
        losses.append(float(l.detach()))
    np.testing.assert_allclose(losses, g["adam/losses"], rtol=1e-3)
    # size of the 5-step update per tensor. The SE fc gradients are sums of nearly cancelling terms whose
    # SIGN moves with 1e-3 noise, and Adam's update is sign-like, so that tensor only gets a loose bound.
    start = cf.fill_state_dict_random(iu.UNet(2, 2, True).state_dict(), seed=7)
    for k, tol in (("outc.weight", 0.05), ("conv4.double_conv.3.weight", 0.05), ("down4.1.double_conv.3.weight", 0.05),
                   ("up1.weight", 0.05), ("down2.1.double_conv.4.weight", 0.05), ("inc.double_conv.6.fc.0.weight", 0.3)):
        nrm = float(g[f"adam/delta/{k}/norm"])
        delta = net.state_dict()[k].cpu() - start[k]
        assert abs(float(delta.norm()) - nrm) / nrm <= tol, k


_BLOCK_NAMES = ["inc", "down1.1", "down2.1", "down3.1", "down4.1"], ["conv1", "conv2", "conv3", "conv4"]


@pytest.mark.parametrize("fixture,shape", [
    ((7, 11, 13, 0.05), (2, 2, 64, 64)),          # the G3r inputs
    ((21, 22, 23, 0.0), (2, 2, 64, 64)),
    ((31, 32, 33, 0.02), (3, 2, 48, 80)),         # ragged
    ("closed-form", (1, 2, 256, 256)),            # the G3 b1_256_train inputs (smooth, many near-zero z)
    ((7, 11, 13, 0.05, "bf16"), (2, 2, 64, 64)),  # the bf16 compute path: same statement at bf16 resolution
    ((31, 32, 33, 0.02, "bf16"), (3, 2, 48, 80)),
])
def test_unet_fp32_gradients_given_equal_relu_decisions(dev, fixture, shape):
    """The tight gradient statement. ReLU is the only discontinuity of the network: where a pre-activation
    sits within forward rounding noise of 0 (|z| < ~1e-5; a few dozen of the 3.5 M elements) two fp32
    implementations may decide differently, and each such flip moves every upstream gradient by ~2e-3
    (tests/tools/debug_fwd_noise.py: all of this path's flips have |z| < 8e-6). The same holds for the ReLU inside
    the SE bottleneck and for near-ties in a max-pool window. So (1) the decisions of the HIP path may differ
    from a float64 oracle's only at |z| < 1e-4 and in at most 40 places, and (2) under the HIP path's OWN
    decisions (oracle ReLU replaced by y * mask, pooling by a gather at the given indices) every parameter
    gradient matches the float64 oracle to 2e-4 rel-L2 (measured 1e-5 ... 3e-5)."""
    import insar_unet_ca_amd as iu
    bf16 = fixture != "closed-form" and len(fixture) > 4
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16 if bf16 else None)
    hw = (shape[0], shape[2], shape[3])
    if fixture == "closed-form":
        net.load_state_dict(cf.fill_state_dict(net.state_dict()))
        x, tgt = cf.make_input(shape), cf.make_target(hw, ignore_every=13)
    else:
        net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=fixture[0]))
        x = cf.make_input_random(shape, seed=fixture[1])
        tgt = cf.make_target_random(hw, seed=fixture[2], ignore_frac=fixture[3])
    net = net.to(dev).train()
    base = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
    iu.CrossEntropyLoss(ignore_index=255)(net(x.to(dev)), tgt.to(dev)).backward()
    plan = net._plan(x.to(dev))
    given_masks, given_idx = {}, {}
    for names, blocks in zip(_BLOCK_NAMES, (plan.enc, plan.dconv)):
        for name, blk in zip(names, blocks):
            given_masks["relu", f"{name}.double_conv.1"] = blk.z1.nchw().cpu() > 0
            # the decision rule of the path: fma(y, scale, shift) > 0 (its exact sign, in float64). The last block's
            # output is never stored (its BN/ReLU/gate pass writes the logits), so the stored activation is not used here.
            u2 = blk.u2
            pre = u2.y.nchw().double() * u2.scale.double().view(1, -1, 1, 1) + u2.shift.double().view(1, -1, 1, 1)
            given_masks["relu", f"{name}.double_conv.4"] = (pre > 0).cpu()
            given_masks["se_relu", f"{name}.double_conv.6"] = blk.se.hid.cpu() > 0
    for i in range(1, 5):       # arg-max of each pool window, from the HIP path's own activations (torch's tie rule)
        given_idx[f"down{i}.0"] = F.max_pool2d(plan.enc[i - 1].out.nchw().cpu(), 2, return_indices=True)[1]

    def oracle_grads(hook):
        work, leaves = OrderedDict(), {}
        for k, v in base.items():
            t = v.double().clone() if v.dtype == torch.float32 else v.clone()
            if orc.is_param(k):
                t.requires_grad_(True)
                leaves[k] = t
            work[k] = t
        orc.DECISION_HOOK = hook
        try:
            orc.cross_entropy(orc.unet_forward(work, x.double(), True, True), tgt).backward()
        finally:
            orc.DECISION_HOOK = None
        return {k: l.grad for k, l in leaves.items()}

    flips, worst_z, pool_flips = 0, 0.0, 0

    def given(kind, name, y):
        nonlocal flips, worst_z, pool_flips
        if kind == "pool":
            idx = given_idx[name]
            pool_flips += int((F.max_pool2d(y.detach(), 2, return_indices=True)[1] != idx).sum())
            return y.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        m = given_masks[kind, name]
        diff = m != (y.detach() > 0)
        flips += int(diff.sum())
        if diff.any():
            worst_z = max(worst_z, float(y.detach()[diff].abs().max()))
        return y * m.to(y.dtype)

    ref = oracle_grads(given)
    # (1) NB: under given decisions upstream pre-activations are those of the HIP path's decisions, so this
    # counts the places where the HIP decision is not the sign of the float64 pre-activation.
    print(f"{fixture} {shape}: {flips} ReLU decisions differ (largest |z| there {worst_z:.2e}), {pool_flips} pool arg-max")
    errs = []
    for k, p in net.named_parameters():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            assert float(p.grad.abs().max()) == 0.0
            continue
        errs.append((rel_l2(p.grad, ref[k]), k))
    errs.sort(reverse=True)
    print("   worst gradient rel-L2:", ", ".join(f"{k} {e:.2e}" for e, k in errs[:6]))
    worst, name = errs[0]
    if bf16:
        # bf16 storage (8 bits of mantissa) moves tens of thousands of decisions (activations within 4e-3 of
        # zero), which is why the comparison is made under the path's own decisions; what remains is bf16
        # rounding of activations, gradients and GEMM operands through ~40 layers against exact float64
        # arithmetic: measured 5-6.5e-2 for every conv / BN / convT tensor, up to 0.2 for the first SE Linear
        # (sums of nearly cancelling per-channel terms).
        for e, k in errs:
            assert e <= (0.35 if k.endswith("fc.0.weight") else 0.1), f"bf16: gradient rel-L2 {e:.3e} at {k}"
        return
    smooth = fixture == "closed-form"      # degenerate (near-constant) channels: invstd up to 316 amplifies the noise
    # closed-form 256x256: measured 425-428 ReLU decisions and 9-10 pool decisions differ: gates at 2x. The largest |z|
    # among them is one position's noise: 1.1e-3 and 2.3e-3 under two summation orders of the SE squeeze fold: gate 4e-3
    assert flips <= (850 if smooth else 40) and worst_z < (4e-3 if smooth else 1e-4) and pool_flips <= 40, \
        f"{flips} ReLU decisions differ, largest |z| there {worst_z:.2e}; {pool_flips} pool decisions differ"
    # generic position: measured 1e-5 ... 2.5e-5, gate 2e-4. Closed-form 256x256 (B=1, near-constant channels whose
    # invstd of up to 316 amplifies rounding noise; torch's own fp32 is off by 0.1 here): measured 5.9e-4 for every conv /
    # BN / convT tensor and 8.0e-3 for the bottleneck's first SE Linear: gates at 2x measured.
    for e, k in errs:
        gate = (1.6e-2 if k.endswith("fc.0.weight") else 1.2e-3) if smooth else 2e-4
        assert e <= gate, f"gradient rel-L2 {e:.3e} at {k} (gate {gate:.1e}; {flips} flips)"


@pytest.mark.parametrize("use_se,cin", [(False, 2), (True, 1)])
def test_unet_variants_against_oracle(dev, use_se, cin):
    """Unet.py-equivalent (use_se=False) and the reference's default in_channels=1."""
    import insar_unet_ca_amd as iu
    net = _unet(dev, use_se=use_se, cin=cin).train()
    sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
    x = cf.make_input((2, cin, 64, 64))
    tgt = cf.make_target((2, 64, 64))
    logits = net(x.to(dev))
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, tgt.to(dev))
    ref = orc.unet_forward(sd, x, use_se=use_se, training=True)
    assert max_rel(logits, ref) <= FWD_TOL
    assert abs(float(loss.detach()) - float(orc.cross_entropy(ref, tgt))) <= 1e-4
    for k, b in net.named_buffers():       # oracle updated `sd` in place
        if not k.endswith("num_batches_tracked"):
            assert max_rel(b, sd[k]) <= FWD_TOL
        else:
            assert int(b) == int(sd[k]) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,ncls,shape", [
    (1, 2, (2, 1, 16, 16)),        # the smallest legal training input: a 1x1 bottleneck with two values per channel
    (3, 3, (5, 3, 32, 16)),        # odd batch, non-square, three classes
    (4, 8, (2, 4, 16, 48)),        # the widest first layer / class count the path takes
    (2, 2, (1, 2, 80, 336)),       # rows longer than a tile; 336 = the widest row the 64->64 kernel's window takes
])
def test_unet_edge_geometries(dev, cin, ncls, shape, dtype):
    """Edge geometries of the whole path (forward, loss, backward, Adam) against the oracle: fp32 within the
    north_star tolerance; bf16 finite, same arg-max on >= 90 % of the pixels and a finite update."""
    import insar_unet_ca_amd as iu
    net = iu.UNet(cin, ncls, True, compute_dtype=dtype)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=41))
    sd = OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items())
    net = net.to(dev).train()
    x = cf.make_input_random(shape, seed=42)
    rng = np.random.Generator(np.random.PCG64(43))
    tgt = torch.from_numpy(rng.integers(0, ncls, size=(shape[0], shape[2], shape[3])).astype(np.int64))
    opt = iu.Adam(net.parameters(), lr=1e-4)
    logits = net(x.to(dev))
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, tgt.to(dev))
    loss.backward()
    opt.step()
    ref = orc.unet_forward(sd, x, use_se=True, training=True)
    assert logits.shape == ref.shape
    if dtype == torch.float32:
        assert max_rel(logits, ref) <= FWD_TOL
        assert abs(float(loss.detach()) - float(orc.cross_entropy(ref, tgt))) <= 1e-4
    else:
        assert (logits.argmax(1).cpu() == ref.argmax(1)).float().mean().item() >= 0.9
    assert all(torch.isfinite(p).all() and torch.isfinite(p.grad).all() for p in net.parameters())


def test_single_value_per_channel_is_refused_like_batchnorm(dev):
    """A 1x1x16x16 training input leaves one value per channel at the bottleneck: nn.BatchNorm2d raises
    ValueError there (and so does the reference); eval mode works."""
    import insar_unet_ca_amd as iu
    net = iu.UNet(1, 2, True).to(dev)
    x = torch.zeros(1, 1, 16, 16, device=dev)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        net.train()(x)
    with torch.no_grad():
        assert net.eval()(x).shape == (1, 2, 16, 16)


# ------------------------------------------------------------------------------------------------
# bf16 compute path
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,shape", [("b2_64_train", (2, 2, 64, 64)), ("b3_48x80_train", (3, 2, 48, 80))])
def test_unet_bf16_against_golden(dev, golden, tag, shape):
    import insar_unet_ca_amd as iu
    g3 = golden("g3_unet")
    net = _unet(dev, dtype=torch.bfloat16).train()
    x = cf.make_input(shape).to(dev)
    tgt = cf.make_target((shape[0], shape[2], shape[3]), ignore_every=13).to(dev)
    logits = net(x)
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, tgt)
    loss.backward()
    ref = torch.from_numpy(g3[f"{tag}/logits/full"]) if f"{tag}/logits/full" in g3.files else None
    if ref is not None:
        err = max_rel(logits, ref)
        agree = (logits.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
        print(f"bf16 vs reference on G3 {tag}: max-rel {err:.3e}, arg-max agreement {agree:.4f}")
        assert err <= BF16_FWD_TOL
        assert agree >= BF16_ARGMAX
    else:
        check_summary(g3, f"{tag}/logits", logits, BF16_FWD_TOL)
    assert abs(float(loss.detach()) - float(g3[f"{tag}/loss"])) <= 2e-2
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    # gradients: only the layers next to the loss are compared (outc sees no ReLU-mask noise); deeper
    # gradients on this 64x64 fixture are dominated by bf16 ReLU-mask flips amplified by the 4x4
    # bottleneck BatchNorm (torch's own fp32-vs-fp64 gap there is already 7-15 %). bf16 training
    # fidelity is gated by the loss-curve test below instead.
    nrm = float(g3[f"{tag}/grad/outc.weight/norm"])
    assert abs(float(net.outc.weight.grad.norm()) - nrm) / nrm <= 0.1


def test_unet_bf16_generic_position(dev, golden):
    """bf16 forward on the generic-position fixture G3r, gated at torch's OWN bf16 error on the same fixture
    (oracle, CPU, measured in-container: all-bf16 8.5e-2 max-rel / 98.2 % arg-max agreement, autocast
    7.0e-2 / 98.4 %; SURVEY 8d's 3e-2 / 99 % probe was taken on torch-default-init weights, which are 3x
    smaller). The HIP bf16 path (bf16 storage, fp32 accumulation and BN statistics) measures 5.8e-2 / 98.6 %."""
    import insar_unet_ca_amd as iu
    g = golden("g3r_unet_random")
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
    net = net.to(dev).train()
    logits = net(cf.make_input_random((2, 2, 64, 64), seed=11).to(dev))
    ref = torch.from_numpy(g["b2_64_train/logits/full"]).reshape(2, 2, 64, 64)
    err = max_rel(logits, ref)
    agree = (logits.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
    print(f"bf16 vs reference on G3r: max-rel {err:.3e}, arg-max agreement {agree:.4f}")
    assert err <= 8.5e-2 and agree >= 0.98


def test_unet_bf16_error_growth_per_level(dev):
    """What bounds the bf16 error: the same weights and input (the G3r fixture's) through the fp32 and the bf16 HIP plans,
    activation by activation. bf16 keeps fp32 accumulators and fp32 BatchNorm / SE statistics, so the error of a level
    is the 2^-9 rounding of its stored activations, renormalised by every BatchNorm: it grows by about half a percent to
    one percent per encoder level and stays flat through the decoder (the skip connections re-inject the shallower,
    more accurate features). Measured on MI355X (rel-L2): enc0..4 0.0045 / 0.0099 / 0.017 / 0.027 / 0.039, dec3..1
    0.045 / 0.046 / 0.046, logits 0.049. Gates = 2x those values: a bf16 kernel that drops fp32 accumulation or fp32
    statistics anywhere shows up as a jump at its level."""
    import insar_unet_ca_amd as iu
    x = cf.make_input_random((2, 2, 64, 64), seed=11).to(dev)
    acts = {}
    for dt in (torch.float32, torch.bfloat16):
        net = iu.UNet(2, 2, True, compute_dtype=dt)
        net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
        net = net.to(dev).train()
        with torch.no_grad():                   # no lease on the plan: _plan(x) below returns the one that just ran
            logits = net(x)
        plan = net._plan(x)
        rec = {f"enc{l}": plan.enc[l].out.nchw() for l in range(5)}
        rec.update({f"dec{l}": plan.dec[l].nchw() for l in (3, 2, 1)})
        rec["logits"] = logits.detach().clone()
        acts[dt] = rec
    growth = {k: rel_l2(acts[torch.bfloat16][k], acts[torch.float32][k]) for k in acts[torch.float32]}
    print("bf16 vs fp32 rel-L2 per level:", {k: round(v, 4) for k, v in growth.items()})
    gates = {"enc0": 0.009, "enc1": 0.02, "enc2": 0.035, "enc3": 0.055, "enc4": 0.08, "dec3": 0.09, "dec2": 0.095, "dec1": 0.095,
             "logits": 0.1}
    for k, gate in gates.items():
        assert growth[k] <= gate, (k, growth[k], gate)
    assert growth["enc0"] < growth["enc4"]


# ------------------------------------------------------------------------------------------------
# loss / metrics / optimizer entry points
# ------------------------------------------------------------------------------------------------
def test_cross_entropy_golden(dev, golden):
    import insar_unet_ca_amd as iu
    g5 = golden("g5_ce")
    lg = (cf.make_input((2, 2, 16, 16), 0.9) * 3.0).to(dev).requires_grad_(True)
    tgt = cf.make_target((2, 16, 16), ignore_every=5).to(dev)
    loss = iu.CrossEntropyLoss(ignore_index=255)(lg, tgt)
    loss.backward()
    assert abs(float(loss.detach()) - float(g5["loss"])) <= 1e-6
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g5["dlogits"], atol=2e-8)
    lg3 = (cf.make_input((2, 3, 8, 8), 0.1) * 2.0).to(dev).requires_grad_(True)
    loss3 = iu.CrossEntropyLoss(ignore_index=255)(lg3, torch.from_numpy(g5["target3"]).to(dev))
    loss3.backward()
    assert abs(float(loss3.detach()) - float(g5["loss3"])) <= 1e-6
    np.testing.assert_allclose(lg3.grad.cpu().numpy(), g5["dlogits3"], atol=2e-8)


def test_dice_against_oracle_definition(dev):
    import insar_unet_ca_amd as iu
    lg = (cf.make_input((2, 2, 16, 16), 0.9) * 3.0)
    tgt = cf.make_target((2, 16, 16), ignore_every=5)
    a = lg.to(dev).requires_grad_(True)
    d = iu.DiceLoss(ignore_index=255)(a, tgt.to(dev))
    d.backward()
    b = lg.clone().requires_grad_(True)
    r = orc.soft_dice_loss(b, tgt)
    r.backward()
    assert abs(float(d.detach()) - float(r.detach())) <= 1e-6
    assert max_rel(a.grad, b.grad) <= 1e-4


@pytest.mark.parametrize("shape", [(3, 2, 32, 48), (2, 3, 16, 20), (2, 4, 8, 12), (2, 2, 15, 17), (2, 5, 16, 16)],
                         ids=["k2_vec4", "k3_vec4", "k4_vec4", "k2_odd_hw", "k5_generic"])
@pytest.mark.parametrize("wce,wd", [(1.0, 1.0), (0.3, 0.7)])
def test_fused_dice_ce_against_oracle(dev, wce, wd, shape):
    """DiceCELoss (one statistics pass + one gradient pass) == ce_weight*CE + dice_weight*Dice of the oracle: the
    four-pixels-per-thread kernels (K <= 4 classes, H*W a multiple of 4) and the per-pixel ones (any K, any size)."""
    import insar_unet_ca_amd as iu
    lg = cf.make_input(shape, 0.9) * 3.0
    base = cf.make_target((shape[0],) + shape[2:], ignore_every=5)              # {0, 1} and 255 at the ignored pixels
    n, yy, xx = torch.meshgrid(*(torch.arange(d) for d in base.shape), indexing="ij")
    tgt = torch.where(base == 255, base, (xx * 3 + yy * 5 + n * 7) % shape[1])    # every class of K occurs
    a = lg.clone().to(dev).requires_grad_(True)
    crit = iu.DiceCELoss(ignore_index=255, ce_weight=wce, dice_weight=wd)
    d = crit(a, tgt.to(dev))
    d.backward()
    b = lg.clone().requires_grad_(True)
    r = wce * orc.cross_entropy(b, tgt) + wd * orc.soft_dice_loss(b, tgt)
    r.backward()
    assert abs(float(d.detach()) - float(r.detach())) <= 2e-6
    assert max_rel(a.grad, b.grad) <= 1e-4
    sep = wce * iu.CrossEntropyLoss(ignore_index=255)(lg.to(dev), tgt.to(dev)) + wd * iu.DiceLoss(ignore_index=255)(lg.to(dev), tgt.to(dev))
    assert abs(float(d.detach()) - float(sep)) <= 2e-6


@pytest.mark.parametrize("case", ["three_of_four", "all_tie", "class1_absent", "ignore255"])
def test_metrics_counts_kat(dev, golden, case):
    g6 = golden("g6_metrics")
    lg = torch.from_numpy(g6[f"{case}/logits"]).to(dev)
    m = _metrics_from_device(lg, torch.from_numpy(g6[f"{case}/mask"]).to(dev), dev)
    np.testing.assert_allclose([m[k] for k in ("acc", "miou", "mpa", "mf1")], g6[f"{case}/expect"], atol=1e-12)


def test_confusion_counts_exact(dev):
    lg = cf.make_input((3, 2, 32, 32), 0.4)
    tg = cf.make_target((3, 32, 32), ignore_every=7)
    from insar_unet_ca_amd import _lib
    from insar_unet_ca_amd._lib import call, ptr
    counts = torch.zeros(3, 2, dtype=torch.int64, device=dev)
    lgd, tgd = lg.to(dev), tg.to(dev)
    call("insar_confusion", ptr(lgd), ptr(tgd), 3, 2, 1024, 255, ptr(counts), _lib.stream_ptr())
    tp, fp, fn = orc.confusion_counts(lg, tg, 2)
    assert counts.cpu().tolist() == [list(map(int, tp)), list(map(int, fp)), list(map(int, fn))]


def test_adam_kernel_against_oracle(dev):
    import insar_unet_ca_amd as iu
    shapes = [(7,), (64, 3, 3, 3), (1000, 33), (5, 4), (70001,)]
    ps = [torch.nn.Parameter(cf.fill_tensor("weight", s, i).to(dev)) for i, s in enumerate(shapes)]
    rs = [p.detach().cpu().clone() for p in ps]
    opt = iu.Adam(ps, lr=1e-3)
    state = {}
    for step in range(4):
        gl = [cf.make_grad(s, 0.1 * step + 0.3 * i) for i, s in enumerate(shapes)]
        for p, g in zip(ps, gl):
            p.grad = g.to(dev)
        v0 = ps[0]._version
        opt.step()
        assert ps[0]._version > v0                  # weight caches key on the tensor version
        orc.adam_update(rs, gl, state, lr=1e-3)
    for p, r in zip(ps, rs):
        assert max_rel(p, r) <= 1e-6
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 4.0 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def test_five_adam_steps_golden(dev, golden):
    """G4: loss curve of 5 training steps of the reference (CE + Adam(lr=1e-4)) from identical init/batches."""
    import insar_unet_ca_amd as iu
    g4 = golden("g4_adam")
    net = _unet(dev).train()
    start = {k: v.detach().clone() for k, v in net.state_dict().items()}
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-4)
    losses = []
    for step in range(5):
        x = cf.make_input((2, 2, 64, 64), salt=0.37 * step).to(dev)
        tgt = cf.make_target((2, 64, 64)).to(dev)
        opt.zero_grad()
        loss = crit(net(x), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g4["losses"], rtol=1e-3)
    # parameter drift: Adam moves every weight by ~lr per step in the direction of sign(grad)
    for k in ("outc.weight", "conv4.double_conv.3.weight", "down4.1.double_conv.3.weight", "inc.double_conv.0.weight",
              "up1.weight", "inc.double_conv.6.fc.0.weight"):
        delta = net.state_dict()[k] - start[k]
        nrm = float(g4[f"delta/{k}/norm"])
        assert abs(float(delta.norm()) - nrm) / nrm <= 0.1, k
    # Running statistics are pinned after ONE step (test_unet_golden_fp32, 1e-3). After five Adam steps
    # they are not a usable parity target on this fixture: the CPU oracle itself moves them by up to 56 %
    # of their range when its input is perturbed by 1e-6 (Adam steps along sign(grad), and the sign of a
    # near-zero gradient entry is rounding noise) - measured in-container, see DESIGN.md.


def test_bf16_training_curve_tracks_fp32_reference(dev, golden):
    """bf16 compute: the 5-step loss curve of the reference (fp32) is tracked within 2 %."""
    import insar_unet_ca_amd as iu
    g4 = golden("g4_adam")
    net = _unet(dev, dtype=torch.bfloat16).train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-4)
    losses = []
    for step in range(5):
        x = cf.make_input((2, 2, 64, 64), salt=0.37 * step).to(dev)
        tgt = cf.make_target((2, 64, 64)).to(dev)
        opt.zero_grad()
        loss = crit(net(x), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g4["losses"], rtol=2e-2)


# ------------------------------------------------------------------------------------------------
# full BASELINE configuration (config 2: bf16, 16 x 2 x 256 x 256): size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_split_backward_coefficients_are_bitwise_the_fused_launch(dev, dtype, monkeypatch):
    """Per-image stage + k1/k2 folded inside the apply pass + batch fold on the side stream (the default) gives the
    same gradient bits as insar_bnse_bwd_coef followed by insar_bnrelu_bwd_apply on the dgrad chain."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    monkeypatch.setattr(engine, "COEF_SIMPLE", False)      # the split schedule is a variant of the two-stage kernels
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(3, 3, 48)
    x, y = x.to(dev), y.to(dev)
    grads = []
    for split in (True, False):
        monkeypatch.setattr(engine, "SPLIT_COEF", split)
        torch.manual_seed(5)
        net = iu.UNet(2, 2, True, compute_dtype=dtype).to(dev).train()
        loss = iu.DiceCELoss(ignore_index=255)(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        grads.append([p.grad.clone() for p in net.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_recomputed_outc_gradient_is_bitwise_the_materialised_one(dev, dtype, monkeypatch):
    """The unit below outc recomputes its incoming gradient from dlogits inside its BatchNorm-backward passes
    (insar_bnrelu_bwd_reduce_outc / _apply_outc, insar_conv1x1_out_wgrad): same bits as insar_conv1x1_out_bwd writing
    the 64-channel tensor and the plain passes reading it, for every parameter gradient."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(7, 3, 48)
    x, y = x.to(dev), y.to(dev)
    grads, outs = [], []
    monkeypatch.setattr(engine, "OUTC_WGRAD_FUSE", False)      # outc's own weight gradient in its own pass: same summation order
    for fuse in (True, False):
        monkeypatch.setattr(engine, "OUTC_FUSE", fuse)
        torch.manual_seed(11)
        net = iu.UNet(2, 2, True, compute_dtype=dtype).to(dev).train()
        logits = net(x)                                  # fused: written by the last unit's BN/ReLU/gate pass itself
        loss = iu.DiceCELoss(ignore_index=255)(logits, y)
        loss.backward()
        torch.cuda.synchronize()
        grads.append([p.grad.clone() for p in net.parameters()])
        net.eval()
        with torch.no_grad():
            outs.append((logits.detach().clone(), net(x).clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("size", [48, 256])
def test_outc_weight_gradient_from_the_reduce_pass(dev, dtype, size, monkeypatch):
    """outc's weight / bias gradient written by the last unit's BatchNorm-backward reduce pass (one read of y for both)
    against the stand-alone pass over y: every other gradient bitwise equal, outc's own to fp32 summation order."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(7, 3 if size == 48 else 2, size)
    x, y = x.to(dev), y.to(dev)
    grads = []
    for fuse in (True, False):
        monkeypatch.setattr(engine, "OUTC_WGRAD_FUSE", fuse)
        torch.manual_seed(11)
        net = iu.UNet(2, 2, True, compute_dtype=dtype).to(dev).train()
        calls = []
        orig = engine.call
        monkeypatch.setattr(engine, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        loss = iu.DiceCELoss(ignore_index=255)(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(engine, "call", orig)
        assert ("insar_conv1x1_out_wgrad_y" in calls) == (not fuse)
        grads.append({n: p.grad.clone() for n, p in net.named_parameters()})
    for n in grads[0]:
        if n.startswith("outc."):
            assert max_rel(grads[0][n], grads[1][n]) <= 2e-5, n
            assert float(grads[0][n].abs().max()) > 0
        else:
            assert torch.equal(grads[0][n], grads[1][n]), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_single_launch_coefficient_stages_are_bitwise_the_two_launches(dev, dtype, monkeypatch):
    """insar_bnse_bwd_coef_fused (units without an SE gate: stage 2 by the work-group that draws the last ticket after
    stage 1; hand-off data written through to memory and read past the caches) against insar_bnse_bwd_coef's two
    launches: every gradient bit for bit, 30 backward passes in a row through the same ticket words (they must reset
    themselves)."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(4, 16, 64)
    x, y = x.to(dev), y.to(dev)
    ref = None
    monkeypatch.setattr(engine, "COEF_SIMPLE", False)          # this test is about the ticket kernel
    for fuse, reps in ((False, 1), (True, 30)):
        monkeypatch.setattr(engine, "COEF_FUSE", fuse)
        torch.manual_seed(5)
        net = iu.UNet(2, 2, True, compute_dtype=dtype).to(dev).train()
        crit = iu.DiceCELoss(ignore_index=255)
        names = [n for n, _ in net.named_parameters()]
        for rep in range(reps):
            for p in net.parameters():
                p.grad = None
            loss = crit(net(x), y)
            loss.backward()
            torch.cuda.synchronize()
            g = [p.grad.clone() for p in net.parameters()]
            if ref is None:
                ref = g
            else:
                bad = [n for n, a, b in zip(names, g, ref) if not torch.equal(a, b)]
                assert not bad, (rep, bad[:6])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_channel_parallel_coefficient_kernel_against_the_two_stage_kernels(dev, dtype, monkeypatch):
    """insar_bn_bwd_coef (units without an SE gate: k1, k2, dgamma, dbeta from all slab rows in one channel-parallel
    launch) against the per-image stage + batch fold: the same sums in another order. fp32: every gradient within 2e-5
    rel-L2; bf16: k1 / k2 differ in the last bits, which moves a few roundings of dy: within a quarter of the distance
    between the bf16 and the fp32 gradient of that tensor + 1e-3 of its norm (the yardstick of tests/test_bstat_gpu.py)."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = (t.to(dev) for t in make_batch(4, 8, 64))

    def run(simple, dt):
        monkeypatch.setattr(engine, "COEF_SIMPLE", simple)
        torch.manual_seed(5)
        net = iu.UNet(2, 2, True, compute_dtype=dt).to(dev).train()
        loss = iu.DiceCELoss(ignore_index=255)(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}

    two, one = run(False, dtype), run(True, dtype)
    assert two[0] == one[0]
    floor = run(False, torch.float32)[1] if dtype == torch.bfloat16 else None
    bad = []
    for n, g in two[1].items():
        den = max(g.norm().item(), 1e-30)
        err = (one[1][n] - g).norm().item()
        allowed = 2e-5 * den if floor is None else 0.25 * (g - floor[n]).norm().item() + 1e-3 * den
        if err > allowed:
            bad.append((n, err / den, allowed / den))
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_gradient_inside_bn_backward_is_bitwise_maxpool2_bwd(dev, dtype, monkeypatch):
    """Arg-max map written by the forward apply+pool pass, pooled gradient added to the skip gradient on the fly in the
    encoder blocks' BatchNorm-backward passes: same gradient bits as insar_maxpool2_bwd accumulating into the skip
    gradient buffer first (the inputs contain exact ties: bf16 activations, ReLU zeros)."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(5, 3, 48)
    x, y = x.to(dev), y.to(dev)
    grads = []
    for fuse in (True, False):
        monkeypatch.setattr(engine, "POOL_FUSE", fuse)
        torch.manual_seed(3)
        net = iu.UNet(2, 2, True, compute_dtype=dtype).to(dev).train()
        loss = iu.DiceCELoss(ignore_index=255)(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        plan = net._plan(x)
        assert (plan.enc[0].pool_arg is not None) == fuse
        grads.append([p.grad.clone() for p in net.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


def test_first_layer_weight_gradient_with_the_apply_pass_inside_is_bitwise(dev, monkeypatch):
    """insar_conv3x3_small_wgrad_fused (BatchNorm / ReLU backward of the first unit evaluated on the way into the weight
    gradient's LDS tile, dy never written) against the apply pass + insar_conv3x3_small_wgrad: same gradient bits
    (bf16, W % 64 == 0: the shapes the fused kernel takes; other shapes keep the two launches)."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(7, 3, 64)
    x, y = x.to(dev), y.to(dev)
    grads = []
    for fuse in (True, False):
        monkeypatch.setattr(engine, "SMALL_WGRAD_FUSE", fuse)
        torch.manual_seed(5)
        net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
        calls = []
        orig = engine.call
        monkeypatch.setattr(engine, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        loss = iu.DiceCELoss(ignore_index=255)(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(engine, "call", orig)
        assert calls.count("insar_conv3x3_small_wgrad_fused") == (1 if fuse else 0)
        assert calls.count("insar_conv3x3_small_wgrad") == (0 if fuse else 1)
        assert (net._plan(x).enc[0].u1.dy is None) == fuse           # dy of the first unit is not even allocated
        grads.append([p.grad.clone() for p in net.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


def test_step_reproducible_over_many_runs_with_side_stream(dev):
    """Race screen for the two-stream step (weight gradients beside the dgrad chain): 150 repeats of fwd+bwd on
    fixed weights must give ONE set of gradient bits. (Regression: a wave passed the K-step barrier of the 64x64
    weight-gradient kernel with LDS reads still queued and ~1e-3 of the launches picked up one fragment of the
    NEXT slab; csrc/common.h, dma_drain_and_barrier.)"""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.DiceCELoss(ignore_index=255)
    x, y = make_batch(0, 16, 256)
    x, y = x.to(dev), y.to(dev)
    plan = net._plan(x)
    seen = set()
    for _ in range(150):
        for p in net.parameters():
            p.grad = None
        loss = crit(net(x), y)
        loss.backward()
        flat = plan.sink.flat()
        seen.add((int(flat.view(torch.int32).to(torch.int64).sum()), float(loss)))
    assert len(seen) == 1, f"{len(seen)} distinct gradient checksums over 150 identical steps"


def test_full_size_properties_bf16(dev):
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.DiceCELoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-4)
    x, y = make_batch(0, 16, 256)
    x, y = x.to(dev), y.to(dev)

    def one_step():
        opt.zero_grad()
        logits = net(x)
        loss = crit(logits, y)
        loss.backward()
        return logits, loss

    l1, loss1 = one_step()
    g1 = [p.grad.clone() for p in net.parameters()]
    bn_before = net.inc.double_conv[1].running_mean.clone()
    l2, loss2 = one_step()
    # (1) no atomics anywhere: the whole step is bitwise reproducible
    assert torch.equal(l1, l2) and float(loss1) == float(loss2)
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))
    assert not torch.equal(bn_before, net.inc.double_conv[1].running_mean)      # running stats did move
    plan = net._plan(x)
    # (2) halos of every activation / gradient buffer are still exactly zero
    for a in [plan.xin, plan.x5] + plan.cat + plan.pooled + plan.dec + plan.dcat + plan.dpooled + plan.ddec:
        assert _halo_abs(a) == 0.0
    # (3) training-mode BN output of the first unit is normalised: z = relu(gamma*xhat + beta) with the
    #     default gamma=1, beta=0 => pre-ReLU mean 0 / var 1 per channel
    u = plan.enc[0].u1
    zpre = u.y.nchw() * u.scale.view(1, -1, 1, 1) + u.shift.view(1, -1, 1, 1)
    assert float(zpre.mean((0, 2, 3)).abs().max()) < 2e-2
    assert float((zpre.var((0, 2, 3), unbiased=False) - 1).abs().max()) < 3e-2
    # (4) SE gates are sigmoids; softmax-CE gradient sums to ~0 over classes
    assert 0.0 < float(plan.enc[0].se.gate.min()) and float(plan.enc[0].se.gate.max()) < 1.0
    # (5) gradients finite, pre-BN conv biases exactly zero, and the loss goes down when we train
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    assert float(net.inc.double_conv[0].bias.grad.abs().max()) == 0.0
    first = float(loss2)
    for _ in range(8):
        _, loss = one_step()
        opt.step()
    assert float(loss) < first


def test_errors_on_device(dev):
    import insar_unet_ca_amd as iu
    net = iu.UNet(2, 2, True).to(dev)
    with pytest.raises(iu.InsarError, match="at least 16"):
        net(torch.zeros(1, 2, 12, 40, device=dev))
    with pytest.raises(iu.InsarError):
        iu.DoubleConv(48, 64).to(dev)(torch.zeros(1, 48, 16, 16, device=dev))
    with pytest.raises(iu.InsarError, match="input tensor"):
        net(torch.zeros(1, 2, 32, 32, device=dev, requires_grad=True))
    # forward under no_grad twice, then a normal step: plans are reusable
    with torch.no_grad():
        a = net.eval()(torch.zeros(1, 2, 32, 32, device=dev))
        b = net(torch.zeros(1, 2, 32, 32, device=dev))
    assert torch.equal(a, b)
    # a frozen parameter gets no gradient (AccumulateGrad semantics); a sub-module in another BatchNorm mode is refused
    net.train()
    net.outc.bias.requires_grad_(False)
    iu.CrossEntropyLoss(ignore_index=255)(net(torch.zeros(2, 2, 32, 32, device=dev)), torch.zeros(2, 32, 32, dtype=torch.long, device=dev)).backward()
    assert net.outc.bias.grad is None and net.outc.weight.grad is not None
    net.down1.eval()
    with pytest.raises(iu.InsarError, match="mixed BatchNorm modes"):
        net(torch.zeros(2, 2, 32, 32, device=dev))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 2, 40, 56), (1, 2, 100, 100), (2, 2, 33, 47)])
def test_tile_sizes_that_are_not_multiples_of_16(dev, shape, dtype):
    """The reference's fallback for such sizes (Unet-ChannalAttention.py:138-139,144-145,150-151,156-157): MaxPool2d(2)
    floors, the transposed conv comes out one pixel short of its skip and is resized bilinearly to it. Against the oracle's
    restatement of that branch (parity unpinned there: the reference's F_T.resize is torchvision's, absent here): logits,
    loss, BatchNorm buffers, and every parameter gradient by norm; the layers next to the loss tightly."""
    import insar_unet_ca_amd as iu
    net = iu.UNet(2, 2, True, compute_dtype=dtype)
    sd = cf.fill_state_dict_random(net.state_dict(), seed=7)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    x = cf.make_input_random(shape, seed=21)
    tgt = cf.make_target_random((shape[0], shape[2], shape[3]), seed=22, ignore_frac=0.03)
    names = [k for k in sd if orc.is_param(k)]
    work = OrderedDict((k, v.clone()) for k, v in sd.items())
    leaves = []
    for k in names:
        work[k] = work[k].requires_grad_(True)
        leaves.append(work[k])
    ref = orc.unet_forward(work, x, use_se=True, training=True)
    ref_loss = orc.cross_entropy(ref, tgt)
    grads = dict(zip(names, torch.autograd.grad(ref_loss, leaves)))
    logits = net(x.to(dev))
    assert logits.shape == ref.shape
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, tgt.to(dev))
    loss.backward()
    fp32 = dtype == torch.float32
    err = max_rel(logits, ref)
    print(f"{shape} {dtype}: logits max-rel {err:.3e}, loss {float(loss.detach()):.6f} vs {float(ref_loss):.6f}")
    assert err <= (FWD_TOL if fp32 else BF16_FWD_TOL)
    assert abs(float(loss.detach()) - float(ref_loss)) <= (1e-4 if fp32 else 2e-2)
    if not fp32:
        assert all(torch.isfinite(p.grad).all() for p in net.parameters())
        return
    for k, b in net.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert max_rel(b, work[k]) <= 1e-4, k
    got = dict(net.named_parameters())
    for k in names:
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            assert float(got[k].grad.abs().max()) == 0.0
            continue
        nrm = float(grads[k].double().norm())
        assert abs(float(got[k].grad.double().norm()) - nrm) <= 5e-2 * nrm, k
    for k, tol in (("outc.weight", 5e-3), ("outc.bias", 5e-3), ("up4.weight", 2e-2), ("up4.bias", 2e-2), ("up1.bias", 2e-2)):
        assert rel_l2(got[k].grad, grads[k]) <= tol, (k, rel_l2(got[k].grad, grads[k]))      # one ReLU flip moves upstream gradients by ~2e-3


def test_train_and_validate_loop_mirrors_reference(dev, golden, tmp_path):
    """train_model / validate_model / compute_metrics with the reference's signatures and history keys
    (Unet-ChannalAttention.py:215-399), metrics read back once per epoch."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import SyntheticTiles
    from insar_unet_ca_amd.train import compute_metrics, train_model, validate_model
    g6 = golden("g6_metrics")
    for case in ("three_of_four", "all_tie", "class1_absent", "ignore255"):
        m = compute_metrics(torch.from_numpy(g6[f"{case}/logits"]).to(dev), torch.from_numpy(g6[f"{case}/mask"]).to(dev), 2)
        np.testing.assert_allclose([m[k] for k in ("acc", "miou", "mpa", "mf1")], g6[f"{case}/expect"], atol=1e-12)
    torch.manual_seed(0)
    net = iu.UNet(2, 2, True)
    train_dl = torch.utils.data.DataLoader(SyntheticTiles(16, 32), batch_size=8, shuffle=False)
    val_dl = torch.utils.data.DataLoader(SyntheticTiles(8, 32, heldout=True), batch_size=8, shuffle=False)
    path = str(tmp_path / "ckpt" / "best.pth")
    hist = train_model(net, train_dl, val_dl, iu.CrossEntropyLoss(ignore_index=255), iu.Adam(net.parameters(), lr=1e-3),
                       dev, num_epochs=2, model_save_path=path, verbose=False)
    assert len(hist) == 2
    assert set(hist[0]) == {"epoch", "train_loss", "train_acc", "train_miou", "train_mpa", "train_mf1",
                            "val_loss", "val_acc", "val_miou", "val_mpa", "val_mf1"}
    assert hist[1]["train_loss"] < hist[0]["train_loss"]
    assert net.training                                  # validate_model restores train mode (:316)
    # the checkpoint is a plain state_dict with the reference's 154 keys and loads into a fresh model
    sd = torch.load(path)
    assert len(sd) == 154
    fresh = iu.UNet(2, 2, True)
    fresh.load_state_dict(sd)
    # validation metrics agree with the oracle's compute_metrics on the same logits
    net.eval()
    x, y = next(iter(val_dl))
    with torch.no_grad():
        lg = net(x.to(dev))
    ours = compute_metrics(lg, y.to(dev), 2)
    ref = orc.compute_metrics(lg.cpu(), y, 2)
    np.testing.assert_allclose([ours[k] for k in ref], [ref[k] for k in ref], atol=1e-12)
    v = validate_model(net, val_dl, iu.CrossEntropyLoss(ignore_index=255), dev, verbose=False)
    assert abs(v["val_miou"] - ours["miou"]) < 1e-12     # one batch: weighted mean == the batch value


def test_train_loop_over_the_voc_reader(dev, tmp_path):
    """End to end on the reference's on-disk format (8f rank 2): VOC-layout tiles -> VOCSegDataset ->
    sharded loader -> train_model on the HIP path with the reference's in_channels=1 model (:464)."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_tile
    root = str(tmp_path / "voc")
    for d in ("JPEGImages", "SegmentationClass", "ImageSets/Segmentation"):
        (tmp_path / "voc" / d).mkdir(parents=True)
    ids = [f"tile_{i:03d}" for i in range(12)]
    for i, name in enumerate(ids):
        img, lab = make_tile(1000 + i, 32, channels=1)
        Image.fromarray(((img[0] * 0.5 + 0.5) * 255).astype(np.uint8), "L").save(f"{root}/JPEGImages/{name}.jpg", quality=95)
        Image.fromarray((lab * 255).astype(np.uint8), "L").save(f"{root}/SegmentationClass/{name}.png")
    open(f"{root}/ImageSets/Segmentation/train.txt", "w").write("\n".join(ids[:8]) + "\n")
    open(f"{root}/ImageSets/Segmentation/val.txt", "w").write("\n".join(ids[8:]) + "\n")
    train_dl = iu.make_loader(iu.VOCSegDataset(root, 32, "train"), batch_size=4, shuffle=True, seed=1)
    val_dl = iu.make_loader(iu.VOCSegDataset(root, 32, "val"), batch_size=4, shuffle=False)
    torch.manual_seed(0)
    net = iu.UNet(in_channels=1, num_classes=2, use_se=True)
    hist = iu.train_model(net, train_dl, val_dl, iu.CrossEntropyLoss(ignore_index=255), iu.Adam(net.parameters(), lr=1e-3),
                          dev, num_epochs=2, model_save_path=str(tmp_path / "best.pth"), verbose=False)
    assert len(hist) == 2 and hist[1]["train_loss"] < hist[0]["train_loss"]
    assert all(np.isfinite(v) for rec in hist for v in rec.values())
    iu.save_history(hist, str(tmp_path / "metrics" / "history.json"))
    assert (tmp_path / "best.pth").exists() and (tmp_path / "metrics" / "history.json").exists()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_miou_within_0p3_pt_of_the_reference(dev, dtype):
    """north_star: "mIoU within 0.3 pt of the reference on identical splits".
    tests/golden/g8c_miou_reference.json holds the held-out mIoU that the IMPORTED REFERENCE (its own UNet, train_model,
    validate_model and compute_metrics, Unet-ChannalAttention.py:100-163, 215-399; torch CPU fp32, build container,
    tests/tools/miou_experiment.py --side reference) reaches on the synthetic "bowl" task — 256 training tiles of 64 x 64,
    batch 8, CE, Adam(lr 1e-4), 30 epochs, evaluated on 1024 held-out tiles — for several seeds; a seed fixes the initial
    weights and the batch order. The HIP path runs the same protocol (insar_unet_ca_amd.train) with the same seeds.
    The task converges (mIoU ~0.98, seed-to-seed std 0.13 pt), so tenths of a point resolve: the gate is that the 95 %
    confidence interval of the PAIRED mean difference (Student t over the seeds) lies inside +-0.3 pt."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "g8c_miou_reference.json")))
    ref_by_seed = {r["seed"]: r["final"]["val_miou"] for r in ref["runs"]}
    seeds = sorted(ref_by_seed)[:5]
    assert len(seeds) >= 5 and ref["config"]["side"] == "reference"
    out = os.path.join(root, "gpurun_out", f"g8c_miou_hip_{dtype}_test.json")
    cfg = ref["config"]
    cmd = [sys.executable, os.path.join(root, "tests", "tools", "miou_experiment.py"), "--side", "hip", "--dtype", dtype,
           "--seeds", ",".join(str(s) for s in seeds), "--out", out]
    for k in ("size", "train", "val", "heldout", "batch", "epochs", "lr", "task"):
        cmd += [f"--{k}", str(cfg[k])]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    got = {r["seed"]: r["final"]["val_miou"] for r in json.load(open(out))["runs"]}
    d = np.array([got[s] - ref_by_seed[s] for s in seeds])
    t975 = {5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365}[len(seeds)]
    mean, ci = float(d.mean()), float(t975 * d.std(ddof=1) / np.sqrt(len(d)))
    r_mean, h_mean = float(np.mean([ref_by_seed[s] for s in seeds])), float(np.mean([got[s] for s in seeds]))
    print(f"{dtype}: held-out mIoU reference {100 * r_mean:.2f} %, HIP {100 * h_mean:.2f} %; paired difference "
          f"{100 * mean:+.3f} pt, 95 % CI +-{100 * ci:.3f} pt over seeds {seeds}")
    assert r_mean >= 0.8                                        # the task is learnt (verdict: mIoU >= 0.8)
    assert abs(mean) + ci <= 0.003, (mean, ci)                  # the whole interval inside +-0.3 pt


def test_boundary_input_forms(dev):
    """forward takes contiguous or channels_last [B,Cin,H,W], fp32 or bf16 tensors; the compute
    dtype follows `compute_dtype`, `set_compute_dtype` or an enclosing torch.autocast; logits are float32."""
    import insar_unet_ca_amd as iu
    net = iu.UNet(2, 2, True)
    net.load_state_dict(cf.fill_state_dict_random(net.state_dict(), seed=7))
    net = net.to(dev).eval()
    x = cf.make_input_random((2, 2, 32, 48), seed=3).to(dev)
    with torch.no_grad():
        base = net(x)
        assert base.dtype == torch.float32 and base.shape == (2, 2, 32, 48)
        assert torch.equal(net(x.contiguous(memory_format=torch.channels_last)), base)
        xb = x.to(torch.bfloat16)
        assert torch.equal(net(xb), net(xb.float()))                   # a bf16 tensor is just its values
        with torch.autocast("cuda", dtype=torch.bfloat16):
            auto = net(x)
        net.set_compute_dtype(torch.bfloat16)
        explicit = net(x)
        net.set_compute_dtype(None)
        again = net(x)
    assert auto.dtype == torch.float32 and torch.equal(auto, explicit)  # autocast == compute_dtype=bf16
    assert torch.equal(again, base)                                     # and back to fp32, bit for bit
    assert 0 < max_rel(explicit, base) < 0.1
