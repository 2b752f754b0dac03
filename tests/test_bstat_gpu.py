"""GPU: BatchNorm-backward sums taken in the epilogue of the GEMM that produces a unit's incoming gradient (InsarBstat,
csrc/igemm.hip / conv3x3_flat.hip / conv3x3_c64.hip) against the pass of their own they replace
(insar_bnrelu_bwd_reduce over (dout, y); Unet-ChannalAttention.py:82-83,85-86 inside loss.backward(), :345).

Kernel level: the same stored gradient and the same y go through both; the folded sums agree to fp32 summation order.
Network level: every parameter gradient of a training step with the fusion on against the fusion off (no decision is
involved in backward — the ReLU masks come from y — so fp32 agrees to summation order and bf16 to a few roundings of dy)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from insar_unet_ca_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand_act(B, H, W, Cn, dtype, dev, seed, scale=1.0):
    from insar_unet_ca_amd import engine
    a = engine.Act.alloc(B, H, W, Cn, dtype, dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    a.buf[:, 1:-1, 1:-1] = (torch.randn(B, H, W, Cn, generator=g) * scale).to(dtype).to(dev)
    return a


def _reference_sums(dout, y, scale, shift, dev):
    """insar_bnrelu_bwd_reduce over (dout, y), folded over its row parts: [2][C] in float64."""
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call, ptr, stream_ptr
    rpp = engine._rows_per_part(dout.B, dout.H)
    rows = -(-dout.H // rpp)
    part = torch.zeros(dout.B * rows, 2, dout.c_len, device=dev)
    call("insar_bnrelu_bwd_reduce", dout.ref, y.ref, ptr(scale), ptr(shift), ptr(part), 1, rpp, stream_ptr())
    torch.cuda.synchronize()
    return part.double().sum(0), part.double().view(dout.B, rows, 2, dout.c_len).sum(1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,B,hw", [(256, 256, 2, 32), (512, 256, 4, 16), (128, 64, 2, 64), (1024, 512, 16, 16)])
def test_igemm_epilogue_sums_against_the_reduce_pass(dev, dtype, cin, cout, B, hw):
    """conv3x3 input gradient through insar_igemm with InsarBstat: the slab rows are image-major tiles, so the per-image
    fold an SE unit needs is checked too (where the row tiles do not straddle images)."""
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call
    ctx = engine.Ctx(dev, dtype)
    dy = _rand_act(B, hw, hw, cout, dtype, dev, 1)
    y = _rand_act(B, hw, hw, cin, dtype, dev, 2)
    dx = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
    p = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    wd = engine.GemmWeight(ctx, p, "conv3").dgrad()
    scale = torch.randn(cin, device=dev)
    shift = torch.randn(cin, device=dev) * 0.3
    M = B * hw * hw
    rows = call("insar_igemm_num_mtiles", M, cin)
    slab = torch.zeros(rows, 2, cin, device=dev)
    engine._igemm(dy, dx, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0, stats=slab, bstat=(y, scale, shift))
    torch.cuda.synchronize()
    ref_tot, ref_img = _reference_sums(dx, y, scale, shift, dev)
    got = slab.double().sum(0)
    den = ref_tot.abs().max().item()
    assert den > 0
    assert (got - ref_tot).abs().max().item() <= 2e-5 * den
    bm = call("insar_igemm_tile_rows", M, cin)
    if (hw * hw) % bm == 0:
        per_img = slab.double().view(B, rows // B, 2, cin).sum(1)
        assert (per_img - ref_img).abs().max().item() <= 2e-5 * ref_img.abs().max().item()
    # the GEMM output itself is what the plain launch writes, bit for bit
    dx2 = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
    engine._igemm(dy, dx2, wd, cin, hw, hw, 1, engine._TAPS3_DGRAD, 0)
    torch.cuda.synchronize()
    assert torch.equal(dx.buf, dx2.buf)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_input_gradient_sums(dev, dtype):
    """ConvTranspose2d(k2, s2) input gradient (stride-2 taps) with InsarBstat: the producer of every decoder / bottleneck
    block's incoming gradient."""
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call
    ctx = engine.Ctx(dev, dtype)
    B, h, cin, cout = 4, 16, 512, 256
    dout = _rand_act(B, 2 * h, 2 * h, cout, dtype, dev, 3)
    y = _rand_act(B, h, h, cin, dtype, dev, 4)
    dx = engine.Act.alloc(B, h, h, cin, dtype, dev)
    p = torch.nn.Parameter(torch.randn(cin, cout, 2, 2, device=dev) * 0.05)
    wd = engine.GemmWeight(ctx, p, "convT").dgrad()
    scale, shift = torch.randn(cin, device=dev), torch.randn(cin, device=dev) * 0.3
    rows = call("insar_igemm_num_mtiles", B * h * h, cin)
    slab = torch.zeros(rows, 2, cin, device=dev)
    engine._igemm(dout, dx, wd, cin, h, h, 2, engine._TAPS2, 0, stats=slab, bstat=(y, scale, shift))
    torch.cuda.synchronize()
    ref_tot, _ = _reference_sums(dx, y, scale, shift, dev)
    assert (slab.double().sum(0) - ref_tot).abs().max().item() <= 2e-5 * ref_tot.abs().max().item()


@pytest.mark.parametrize("two", [0, 1, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,B,hw", [(128, 128, 16, 64), (64, 128, 4, 128)])
def test_flat_kernel_epilogue_sums_against_the_reduce_pass(dev, dtype, cin, cout, B, hw, two, monkeypatch):
    """two: the two-work-group build of the kernel (csrc/conv3x3_flat2.hip; bf16), persistent (1) and one tile per group (2)."""
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call
    if two and dtype != torch.bfloat16:
        pytest.skip("the two-work-group kernel is bf16 only")
    monkeypatch.setattr(engine, "FLAT2", two)
    ctx = engine.Ctx(dev, dtype)
    dy = _rand_act(B, hw, hw, cout, dtype, dev, 5)
    y = _rand_act(B, hw, hw, cin, dtype, dev, 6)
    if not call("insar_conv3x3_flat_ok", dy.ref, cin):
        pytest.skip("grid too small for the flat kernel")
    dx = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
    p = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    wd = engine.GemmWeight(ctx, p, "conv3").dgrad()
    scale, shift = torch.randn(cin, device=dev), torch.randn(cin, device=dev) * 0.3
    rows = call("insar_conv3x3_flat_stat_rows", dy.ref, cin, engine._flat_flags(1, dy))
    slab = torch.zeros(rows, 2, cin, device=dev)
    engine._conv3x3_flat(dy, dx, wd, 1, slab, bstat=(y, scale, shift))
    torch.cuda.synchronize()
    ref_tot, _ = _reference_sums(dx, y, scale, shift, dev)
    assert (slab.double().sum(0) - ref_tot).abs().max().item() <= 2e-5 * ref_tot.abs().max().item()
    dx2 = engine.Act.alloc(B, hw, hw, cin, dtype, dev)
    engine._conv3x3_flat(dy, dx2, wd, 1, None)
    torch.cuda.synchronize()
    assert torch.equal(dx.buf, dx2.buf)


def test_c64_kernel_epilogue_sums_against_the_reduce_pass(dev):
    """64 -> 64 channels, bf16 (the persistent register-weight kernel of the full-resolution level)."""
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd._lib import call
    dtype = torch.bfloat16
    ctx = engine.Ctx(dev, dtype)
    B, hw = 4, 128
    dy = _rand_act(B, hw, hw, 64, dtype, dev, 8)
    y = _rand_act(B, hw, hw, 64, dtype, dev, 9)
    assert call("insar_conv3x3_c64_ok", dy.ref, 64)
    dx = engine.Act.alloc(B, hw, hw, 64, dtype, dev)
    p = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
    wd = engine.GemmWeight(ctx, p, "conv3").dgrad()
    scale, shift = torch.randn(64, device=dev), torch.randn(64, device=dev) * 0.3
    rows = call("insar_conv3x3_c64_rows", dy.ref)
    slab = torch.zeros(rows, 2, 64, device=dev)
    engine._conv3x3_c64(dy, dx, wd, 1, slab, bstat=(y, scale, shift))
    torch.cuda.synchronize()
    ref_tot, _ = _reference_sums(dx, y, scale, shift, dev)
    assert (slab.double().sum(0) - ref_tot).abs().max().item() <= 2e-5 * ref_tot.abs().max().item()
    dx2 = engine.Act.alloc(B, hw, hw, 64, dtype, dev)
    engine._conv3x3_c64(dy, dx2, wd, 1, None)
    torch.cuda.synchronize()
    assert torch.equal(dx.buf, dx2.buf)


@pytest.mark.parametrize("dtype,size,B", [(torch.float32, 256, 2), (torch.bfloat16, 256, 2), (torch.float32, 64, 4),
                                        (torch.bfloat16, 128, 16)])
def test_training_step_with_and_without_the_fusion(dev, dtype, size, B, monkeypatch):
    """Every parameter gradient and the loss of one training step: sums from the GEMM epilogues against sums from the
    reduce pass. At 256 x 256 the SE units of the 32 x 32 ... 128 x 128 levels are fused too (row tiles within one image);
    at 64 x 64 most of them fall back. The fused run must actually have skipped reduce passes. fp32: 2e-5 rel-L2 per tensor
    (summation order). bf16: the two runs round a few elements of dy differently, which near-cancelling sums (the biases in
    front of a BatchNorm-free path) amplify; the yardstick is bf16's own noise on the same tensor — the distance between
    the unfused bf16 gradient and the fp32 gradient of the same weights: the fusion may move a tensor by at most a
    quarter of that (+ 1e-3 of its norm)."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch
    x, y = (t.to(dev) for t in make_batch(7, B, size))

    def run(fuse, dt):
        monkeypatch.setattr(engine, "BSTAT_FUSE", fuse)
        torch.manual_seed(11)
        net = iu.UNet(2, 2, True, compute_dtype=dt).to(dev).train()
        crit = iu.DiceCELoss(ignore_index=255)
        loss = crit(net(x), y)
        loss.backward()
        torch.cuda.synchronize()
        plan = next(iter(net._plans.plans.values()))[0]
        units = [u for b in plan.enc + plan.dconv for u in (b.u1, b.u2)]
        return (float(loss.detach()), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()},
                sum(1 for u in units if u.bred is not None))

    off, on = run(False, dtype), run(True, dtype)
    assert off[2] == 0 and on[2] >= (9 if size >= 128 else 4), (off[2], on[2])
    assert off[0] == on[0]
    floor = run(False, torch.float32)[1] if dtype == torch.bfloat16 else None
    bad = []
    for n, g in off[1].items():
        den = max(g.norm().item(), 1e-30)
        err = (on[1][n] - g).norm().item()
        allowed = 2e-5 * den if floor is None else 0.25 * (g - floor[n]).norm().item() + 1e-3 * den
        if err > allowed:
            bad.append((n, err / den, allowed / den))
    assert not bad, bad[:8]
