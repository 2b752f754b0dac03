"""CPU-only: the host side of the product (module surface, plans, optimizer state, data, DP
bucketing) without running any kernel."""
import collections

import numpy as np
import pytest
import torch

import insar_unet_ca_amd as iu
from insar_unet_ca_amd import _lib, engine, modules
from insar_unet_ca_amd.data import SyntheticTiles, make_batch, make_tile
from insar_unet_ca_amd.parallel import plan_buckets
from oracle import unet_ca_oracle as orc


def test_state_dict_contract_matches_reference(golden):
    g3 = golden("g3_unet")
    net = iu.UNet(in_channels=2, num_classes=2, use_se=True)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g3["state_dict_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g3["state_dict_shapes"]]
    assert sum(p.numel() for p in net.parameters()) == 31261122
    # Unet.py-equivalent (no attention) and the reference default in_channels=1
    assert list(iu.UNet(1, 2, False).state_dict().keys()) == list(orc.state_dict_template(1, 2, False).keys())


def test_state_dict_roundtrip_with_oracle_template():
    from oracle import closed_form as cf
    net = iu.UNet(2, 2, True)
    filled = cf.fill_state_dict(orc.state_dict_template(2, 2, True))
    net.load_state_dict(filled)                       # strict: same keys + shapes
    for k, v in net.state_dict().items():
        assert torch.equal(v, filled[k])


def test_default_init_consumes_rng_like_the_reference():
    """Same module construction order => same torch default init under the same seed."""
    torch.manual_seed(123)
    a = iu.UNet(2, 2, True).state_dict()
    torch.manual_seed(123)
    b = iu.UNet(2, 2, True).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["inc.double_conv.0.weight"]
    bound = 1.0 / np.sqrt(2 * 9)                      # kaiming_uniform(a=sqrt(5)) bound = 1/sqrt(fan_in)
    assert float(w.abs().max()) <= bound + 1e-6


def test_no_cpu_fallback():
    net = iu.UNet(2, 2, True)
    with pytest.raises(iu.InsarError, match="no CPU fallback"):
        net(torch.zeros(1, 2, 16, 16))
    with pytest.raises(iu.InsarError, match="no CPU fallback"):
        iu.DoubleConv(64, 64)(torch.zeros(1, 64, 16, 16))
    with pytest.raises(iu.InsarError, match="no CPU fallback"):
        iu.CrossEntropyLoss(ignore_index=255)(torch.zeros(1, 2, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(iu.InsarError, match="no CPU fallback"):
        iu.Adam([p]).step()


def test_unsupported_configurations_fail_loudly():
    with pytest.raises(iu.InsarError):
        iu.CrossEntropyLoss(reduction="sum")
    with pytest.raises(iu.InsarError):
        iu.Adam([torch.nn.Parameter(torch.zeros(1))], weight_decay=0.1)


@pytest.fixture
def mocked_abi(monkeypatch):
    calls = []

    def fake_call(name, *a):
        calls.append(name)
        if name == "insar_igemm_num_mtiles":
            return (a[0] + 127) // 128
        if name == "insar_igemm_tile_rows":
            return 128
        if name == "insar_wgrad_tile":
            return 64
        if name == "insar_wgrad_tile_pair":
            return (64 << 16) | 64
        if name == "insar_wgrad_conv3_tile":
            return 0
        if name in ("insar_conv3x3_flat_ok", "insar_conv3x3_flat_num_mtiles"):
            return 0
        if name == "insar_conv3x3_small_fwd_rows":
            return 64
        if name == "insar_conv3x3_small_wgrad_blocks":
            return min(a[0] * a[1], 512)
        if name == "insar_conv1x1_out_bwd_blocks":
            return min(a[0] * a[1], 1024)
        if name == "insar_ce_blocks":
            return min((a[0] + 255) // 256, 1024)
        return 0

    for m in (_lib, engine, modules):
        monkeypatch.setattr(m, "call", fake_call, raising=False)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: 0)
    monkeypatch.setattr(modules, "_require_device", lambda x, who: None)
    return calls


def test_plan_launch_sequence(mocked_abi):
    """The plan issues the expected number of launches per kernel family for one training step."""
    net = iu.UNet(2, 2, True)
    x = torch.zeros(2, 2, 32, 32)
    y = net(x)
    assert y.shape == (2, 2, 32, 32) and y.dtype == torch.float32
    y.sum().backward()
    c = collections.Counter(mocked_abi)
    assert c["insar_igemm"] == 17 + 4 + 17 + 4          # conv fwd (18-1 direct) + convT fwd + dgrads
    assert c["insar_wgrad"] == 17 + 4
    assert c["insar_conv3x3_small_fwd"] == 1 and c["insar_conv3x3_small_wgrad"] == 1
    assert c["insar_bn_finalize"] == 18 and c["insar_se_excite"] == 9
    # pool forward rides on the apply pass (with its arg-max map); pool backward rides on the reduce / apply passes
    assert c["insar_bn_relu_apply_pool_arg"] == 4 and c.get("insar_maxpool2_bwd", 0) == 0
    assert c["insar_bnrelu_bwd_reduce_pool"] == 4 and c["insar_bnrelu_bwd_apply_pool"] == 4
    # the unit that feeds outc recomputes its incoming gradient from dlogits; outc only produces its parameter gradients
    # coefficients: the channel-parallel one-launch kernel for the nine units without an SE gate, two stages for the nine with one
    assert c["insar_bnse_bwd_coef"] == 9 and c["insar_bn_bwd_coef"] == 9 and c.get("insar_bnse_bwd_coef_fused", 0) == 0
    assert c["insar_bnrelu_bwd_apply"] == 13 and c["insar_bnrelu_bwd_apply_outc"] == 1
    # BatchNorm-backward sums from the epilogue of the GEMM that produces the incoming gradient (InsarBstat): all nine first
    # units of a block (no SE gate: any row partition), and of the four SE units fed by a transposed conv's input gradient
    # the one whose 128-row GEMM tiles stay inside an image (16 x 16 pixels here; 4, 16 and 64 pixels do not): 3 plain
    # reduce passes are left, + the 4 bias reductions of the transposed convs on the same entry point
    plan0 = net._plan(x)
    fused = [u.name for b in plan0.enc + plan0.dconv for u in (b.u1, b.u2) if u.bred is not None]
    assert len(fused) == 10 and "conv3.3" in fused and all(n.endswith(".0") for n in fused if n != "conv3.3"), fused
    assert c["insar_bnrelu_bwd_reduce"] == 3 + 4
    # ... and those come out of the same reduce pass (one read of y), folded by a column sum: no pass of their own
    assert c["insar_bnrelu_bwd_reduce_outc"] == 1 and c.get("insar_conv1x1_out_wgrad_y", 0) == 0 and c.get("insar_conv1x1_out_bwd", 0) == 0
    # ... and in forward its BN/ReLU/gate pass writes the logits itself (no 64-channel output tensor, no separate outc launch)
    assert c["insar_bn_relu_apply_outc"] == 1 and c.get("insar_conv1x1_out_fwd", 0) == 0
    assert all(p.grad is not None and p.grad.shape == p.shape for p in net.parameters())
    # gradients alias the plan's flat buffer (no per-step clone) and the next backward must not clobber them
    plan = net._plan(x)
    assert net.outc.weight.grad.data_ptr() == plan.sink.view(net.outc.weight).data_ptr()
    first = plan.sink.active
    net(x).sum().backward()
    assert plan.sink.active != first                    # live .grad aliases buffer 0 -> buffer 1 is used


def test_plan_for_a_tile_size_that_is_not_a_multiple_of_16(mocked_abi):
    """40 x 56: levels 40x56, 20x28, 10x14, 5x7, 2x3. Level 3 is odd: its max-pool floors (stand-alone pool launches instead
    of the fused BatchNorm passes) and up1's 4x6 output is resized to the 5x7 skip (Unet-ChannalAttention.py:138-139)."""
    net = iu.UNet(2, 2, True)
    x = torch.zeros(2, 2, 40, 56)
    y = net(x)
    assert y.shape == (2, 2, 40, 56)
    y.sum().backward()
    c = collections.Counter(mocked_abi)
    assert c["insar_resize_bilinear_fwd"] == 1 and c["insar_resize_bilinear_bwd"] == 1
    assert c["insar_maxpool2_fwd"] == 1 and c["insar_maxpool2_bwd"] == 1
    assert c["insar_bn_relu_apply_pool_arg"] == 3 and c["insar_bnrelu_bwd_reduce_pool"] == 3
    plan = net._plan(x)
    assert [u.resize for u in plan.up] == [True, False, False, False]
    assert (plan.up[0].conv_out.H, plan.up[0].conv_out.W, plan.up[0].out.H, plan.up[0].out.W) == (4, 6, 5, 7)
    assert (plan.x5.H, plan.x5.W) == (2, 3)


def test_plan_rejects_shapes_outside_the_hot_path(mocked_abi):
    net = iu.UNet(2, 2, True)
    with pytest.raises(iu.InsarError, match="at least 16"):
        net(torch.zeros(1, 2, 100, 12))
    with pytest.raises(iu.InsarError, match="input channels"):
        net(torch.zeros(1, 3, 32, 32))


def test_bucket_planning():
    sizes = [10, 10, 100, 5, 5, 200, 1]
    assert plan_buckets(sizes, 50) == [2, 5, 6]
    assert plan_buckets(sizes, 10 ** 9) == [6]
    assert plan_buckets(sizes, 1) == list(range(7))


def test_synthetic_tiles_are_deterministic_and_in_range():
    a, la = make_tile(1234, 64)
    b, lb = make_tile(1234, 64)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    assert a.shape == (2, 64, 64) and a.dtype == np.float32 and la.dtype == np.int64
    assert -1.0 <= a.min() and a.max() <= 1.0
    np.testing.assert_allclose(a[0] ** 2 + a[1] ** 2, 1.0, atol=1e-5)   # (cos, sin) of the wrapped phase
    assert set(np.unique(la)) <= {0, 1}
    x, y = make_batch(0, 3, 32)
    assert x.shape == (3, 2, 32, 32) and y.shape == (3, 32, 32)
    ds = SyntheticTiles(5, 32)
    assert len(ds) == 5 and torch.equal(ds[2][0], make_batch(2, 1, 32)[0][0])
    frac = np.mean([make_tile(1000 + i, 256)[1].mean() for i in range(8)])
    assert 0.005 < frac < 0.12


def test_adam_state_dict_is_torch_compatible():
    p = [torch.nn.Parameter(torch.ones(3))]
    ours, ref = iu.Adam(p, lr=1e-4), torch.optim.Adam(p, lr=1e-4)
    ko = {k for k in ours.state_dict()["param_groups"][0]}
    kr = {k for k in ref.state_dict()["param_groups"][0]}
    assert ko == kr
    ref.load_state_dict(ours.state_dict())


@pytest.mark.parametrize("case", ["three_of_four", "all_tie", "class1_absent", "ignore255"])
def test_metrics_arithmetic_matches_reference_fixture(golden, case):
    """train.metrics_from_counts (host half of compute_metrics) against the reference's own outputs (G6)."""
    from insar_unet_ca_amd.train import metrics_from_counts
    g6 = golden("g6_metrics")
    tp, fp, fn = orc.confusion_counts(torch.from_numpy(g6[f"{case}/logits"]), torch.from_numpy(g6[f"{case}/mask"]), 2)
    m = metrics_from_counts(tp, fp, fn)
    np.testing.assert_allclose([m[k] for k in ("acc", "miou", "mpa", "mf1")], g6[f"{case}/expect"], atol=1e-12)


def test_plan_cache_is_bounded():
    """At most MAX_KEYS geometries stay cached; idle least-recently-used ones go first, busy ones never."""
    cache = modules._PlanCache()

    class P:
        def __init__(self):
            self.busy = False

    made = [cache.get(("k", i), P) for i in range(3)]
    made[0].busy = True                                   # e.g. between a forward and its backward
    for i in range(3, 8):
        cache.get(("k", i), P)
    assert len(cache.plans) <= cache.MAX_KEYS + 1         # the busy geometry may exceed the cap by one
    assert ("k", 0) in cache.plans and ("k", 7) in cache.plans and ("k", 1) not in cache.plans
    assert cache.get(("k", 7), P) is cache.plans[("k", 7)][0]      # idle plan reused, not rebuilt


def test_side_stream_weight_gradients_aim_at_half_the_slots():
    """Split-K factor of the row-of-taps weight gradient: the cost model is asked for half the work-group slots when the
    launch runs beside the dgrad chain (engine.WGRAD_FILL), for all of them when it runs alone; fp32 always fills the chip."""
    from insar_unet_ca_amd import engine, _lib
    ks = 16 * 256 * 256 // 64
    full = engine._wgrad_nsplit(3, ks, 9 * 64 * 64, 64, 64, 2, taps_per_wg=3, fill=1.0)
    half = engine._wgrad_nsplit(3, ks, 9 * 64 * 64, 64, 64, 2, taps_per_wg=3, fill=0.5)
    assert half < full and 0.4 * full <= half <= 0.75 * full          # 256 (the cap) vs 169 on this layer
    assert engine._wgrad_nsplit(192, 64, 9 * 1024 * 1024, 128, 128, 2, taps_per_wg=3, fill=0.5) <= 2     # deep layers: (almost) no split
    class C:                     # the only attribute _side_fill reads
        code = _lib.BF16
    assert engine._side_fill(C, 0.5) == 0.5
    C.code = _lib.F32
    assert engine._side_fill(C, 0.5) == 1.0


def test_c64_ring_never_overwrites_a_live_row():
    """The persistent 64->64 kernel (csrc/conv3x3_c64.hip) prefetches tile t+1's pixels by LDS-DMA while tile t's
    taps are still reading the ring: for every image width the newest prefetched pixel must not share a ring row
    with any pixel of tile t's window. (Round 1 sized the ring 0..A rows short: W % 64 in {32, 48} aliased.)"""
    import ctypes as C
    lib = _lib.load()
    out = (C.c_int32 * 6)()
    checked = 0
    for W in list(range(16, 400, 2)):
        act = _lib.InsarAct(0x1000, 4, 16, W, 64, 0, 64, _lib.BF16, 0)
        ok = lib.insar_conv3x3_c64_geometry(C.byref(act), out)
        assert ok == lib.insar_conv3x3_c64_ok(C.byref(act), 64)
        if not ok:
            continue
        A, Af, o, R, ntiles, T = list(out)
        assert A == W + 3 and Af == (A // 64) * 64 and o == Af - A and R % 64 == 0
        assert R * 128 + 8 * 64 * 2 * 4 <= 160 * 1024
        for t0 in (0, 1, 5):
            first = t0 * T + o - A
            u0 = (first // 64) * 64                      # floor to a multiple of 64 (python // floors negatives too)
            for t in range(t0, t0 + 6):
                lo = t * T + o - A                       # oldest pixel tile t reads
                hi = t * T + T + Af                      # exclusive end of tile t's window (= its fetch frontier)
                pre = hi + T                             # frontier after the four units fetched at the top of tile t
                live = {(p - u0) % R for p in range(lo, hi)}
                new = {(p - u0) % R for p in range(hi, pre)}
                assert len(live) == hi - lo, f"W={W}: tile window wraps onto itself (R={R})"
                assert not (live & new), f"W={W}, tile {t}: {len(live & new)} prefetched rows alias live rows (R={R})"
        checked += 1
    assert checked > 100


def test_deeplab_state_dict_contract_and_tap_logic():
    """Config 5's module tree carries the names / shapes / order of the reference wrapper's state_dict as the oracle
    restates it (726 entries, 364 of them the wrapper's aliases, 39 635 906 parameters = torchvision's
    DeepLabV3-ResNet50 without the aux head, with the 1-channel stem and the 2-class classifier), and the host-side
    geometry of the dilated convolutions (live taps, out-of-bounds flag) is what the layer shapes imply."""
    from insar_unet_ca_amd import deeplab
    from oracle import deeplab_oracle as dlo
    net = iu.DeepLabV3_SingleChannel_Attn(num_classes=2, backbone="resnet50", pretrained=False)
    sd, tmpl = net.state_dict(), dlo.state_dict_template(2)
    assert list(sd.keys()) == list(tmpl.keys()) and len(sd) == 726
    assert all(tuple(sd[k].shape) == tuple(tmpl[k].shape) for k in tmpl)
    assert sum(p.numel() for p in net.parameters()) == 39635906
    assert sd["backbone.layer3.0.downsample.0.weight"].data_ptr() == sd["model.backbone.layer3.0.downsample.0.weight"].data_ptr()
    assert sd["upsample_conv.weight"].shape == (2, 256, 1, 1) and sd["model.backbone.conv1.weight"].shape == (64, 1, 7, 7)
    with pytest.raises(ValueError):
        iu.DeepLabV3_SingleChannel_Attn(backbone="vgg")
    with pytest.raises(iu.InsarError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 64, 64))
    # dilation / stride plan of the 16 bottlenecks (torchvision _make_layer with replace_stride_with_dilation)
    specs = {s[0]: s[1:] for s in dlo.block_specs()}
    assert specs["layer2.0"] == (256, 128, 2, 1, True) and specs["layer3.0"] == (512, 256, 1, 1, True)
    assert specs["layer3.1"][3] == 2 and specs["layer4.0"][3] == 2 and specs["layer4.2"][3] == 4
    # a tap is live when it reaches the interior for some output position
    live = deeplab._live
    assert live(8, 8, 1, 0) and live(8, 8, 1, -7) and not live(8, 8, 1, -8) and not live(8, 8, 1, 12)
    assert live(32, 32, 1, 24) and not live(32, 32, 1, 36) and live(32, 64, 2, -1) and live(32, 64, 2, 1)


def test_deeplab_checkpoint_loaders_follow_the_reference_call_site():
    """(1) a reference checkpoint made with pretrained=True carries model.aux_classifier.* keys the forward pass
    never uses: a strict load ignores them; (2) the reference's pretrained=False call site still starts from an ImageNet
    ResNet-50 backbone: load_backbone_state_dict takes such a state_dict, dropping fc.* and mean-reducing the 3-channel stem
    (DeepLabV3-ChannelAttention.py:105-118)."""
    from collections import OrderedDict
    import torch
    import insar_unet_ca_amd as iu
    net = iu.DeepLabV3_SingleChannel_Attn(2)
    sd = OrderedDict(net.state_dict())
    sd["model.aux_classifier.0.weight"] = torch.zeros(256, 1024, 3, 3)
    sd["model.aux_classifier.4.bias"] = torch.zeros(21)
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    resnet = OrderedDict((k[len("backbone."):], torch.randn_like(v) if v.is_floating_point() else v.clone())
                         for k, v in net.state_dict().items() if k.startswith("backbone."))
    resnet["conv1.weight"] = torch.randn(64, 3, 7, 7)
    resnet["fc.weight"], resnet["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    net.load_backbone_state_dict(resnet)
    assert torch.allclose(net.backbone["conv1"].weight, resnet["conv1.weight"].mean(1, keepdim=True))
    assert torch.equal(net.model.backbone["layer3"][2].conv2.weight, resnet["layer3.2.conv2.weight"])


def test_switch_table_is_read_off_the_source(monkeypatch):
    """insar_unet_ca_amd.switches: every INSAR_* variable the package reads is known with its default; what the environment
    sets differently (or what the package does not know) is reported — bench.py prints it and refuses a default run."""
    import os
    from insar_unet_ca_amd import switches
    known = switches.declared([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")])
    for name, default in (("INSAR_WGRAD_FILL", "0.5"), ("INSAR_FLAT2", "1"), ("INSAR_TAPE", "1"), ("INSAR_FLAT_PERSIST", "2"), ("INSAR_HIP_LIB", None),
                          ("INSAR_GATE_FUSE", "1"), ("INSAR_ADAM_CHUNK", "8192"), ("INSAR_MAIN_PRIORITY", None)):
        assert name in known and known[name] == default, (name, known.get(name))
    for k in list(os.environ):
        if k.startswith("INSAR_"):
            monkeypatch.delenv(k)
    assert switches.non_default() == {}
    monkeypatch.setenv("INSAR_WGRAD_FILL", "0.5")            # the default, spelled out: not a change
    monkeypatch.setenv("INSAR_TAPE", "0")
    monkeypatch.setenv("INSAR_FLAT_ROWS", "0")
    monkeypatch.setenv("INSAR_NO_SUCH_SWITCH", "1")
    nd = switches.non_default()
    assert set(nd) == {"INSAR_TAPE", "INSAR_FLAT_ROWS", "INSAR_NO_SUCH_SWITCH"}
    assert nd["INSAR_TAPE"]["kernel_selecting"] is False and nd["INSAR_FLAT_ROWS"]["kernel_selecting"] is True
    assert nd["INSAR_NO_SUCH_SWITCH"].get("unknown") is True
