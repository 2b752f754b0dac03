"""GPU parity of config 5, DeepLabV3-CA (DeepLabV3-ChannelAttention.py:83-162), against oracle/deeplab_oracle.py.
PARITY UNPINNED for the torchvision part of the network (torchvision is absent, its loader fetches weights);
ChannelAttentionModule itself is pinned by fixture G7 (tests/test_parity_gpu.py). What these tests establish is that
the HIP plan computes the published DeepLabV3-ResNet50 architecture as the oracle restates it: forward <= 1e-3 (fp32),
parameter gradients, BatchNorm buffers, dropout under a given mask, Adam steps, and the bf16 benchmark geometry."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import deeplab_oracle as dlo
from tests.helpers import max_rel, to_np

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from insar_unet_ca_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = to_np(a), to_np(b)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


def _make(dev, seed, dtype=torch.float32, p_drop=None):
    import insar_unet_ca_amd as iu
    torch.manual_seed(seed)
    net = iu.DeepLabV3_SingleChannel_Attn(num_classes=2, backbone="resnet50", pretrained=False, compute_dtype=dtype)
    # generic-position BatchNorm parameters and buffers (the default gamma = 1, beta = 0, mean 0, var 1 hide mistakes)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_var.copy_(1.0 + 0.2 * torch.rand(m.bias.shape, generator=g))
    if p_drop is not None:
        net.aspp.project[3].p = p_drop
    sd = OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items())
    # re-alias the clone like the reference's state_dict (model.backbone.* is backbone.*, ...)
    tmpl = dlo.state_dict_template(2)
    owner = {}
    for k, v in tmpl.items():
        owner.setdefault(v.data_ptr() if v.numel() else id(v), k)
    for k, v in tmpl.items():
        first = owner[v.data_ptr() if v.numel() else id(v)]
        if first != k:
            sd[k] = sd[first]
    return net.to(dev), sd


def _input(shape, seed):
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(seed, shape[0], shape[2], channels=1)
    return x, y


@pytest.mark.parametrize("shape", [(2, 1, 64, 64), (2, 1, 128, 128), (3, 1, 96, 160)])
def test_deeplab_eval_forward_against_oracle(dev, shape):
    net, sd = _make(dev, 11)
    net.eval()
    x, _ = _input(shape, 5) if shape[2] == shape[3] else (torch.randn(shape, generator=torch.Generator().manual_seed(3)).clamp(-1, 1), None)
    with torch.no_grad():
        got = net(x.to(dev))
        ref = dlo.forward(sd, x, training=False)
    assert got.shape == ref.shape == (shape[0], 2, shape[2], shape[3])
    err = max_rel(got, ref)
    print(f"DeepLabV3-CA eval {shape}: logits max-rel {err:.3e}")
    assert err <= FWD_TOL


def _oracle_grads(sd, x, y, dt):
    """logits, loss and parameter gradients of the oracle in precision `dt`; also returns the state it updated."""
    names = [k for k in dlo.primary_keys(sd) if dlo.is_param(k)]
    conv = lambda v: v.to(dt) if v.is_floating_point() else v.clone()
    work = OrderedDict()
    for k in dlo.primary_keys(sd):
        work[k] = conv(sd[k])
    leaves = []
    for k in names:
        work[k] = work[k].clone().requires_grad_(True)
        leaves.append(work[k])
    out = dlo.forward(work, x.to(dt), training=True)
    loss = dlo.cross_entropy(out, y)
    grads = dict(zip(names, torch.autograd.grad(loss, leaves)))
    return out.detach(), float(loss), grads, work


@pytest.mark.parametrize("shape", [(2, 1, 64, 64), (2, 1, 128, 128)])
def test_deeplab_train_forward_backward_against_oracle(dev, shape):
    """Training-mode forward + backward. The reference point is the oracle in FLOAT64; the tolerance of every gradient
    tensor is calibrated by the oracle's own float32-vs-float64 disagreement on the same fixture (small maps + batch
    statistics make ReLU / max-pool decisions chaotic: torch's fp32 gradients are 1.5e-2 rel-L2 off its fp64 ones in the
    median here, measured in the build container): per tensor, HIP fp32 must be within 3x the oracle's fp32 error on that
    tensor or within 2x the oracle's median fp32 error (different implementations flip different decisions), and its
    median error must not exceed twice the oracle's."""
    import insar_unet_ca_amd as iu
    net, sd = _make(dev, 21, p_drop=0.0)
    net.train()
    x, y = _input(shape, 9)
    ref, ref_loss, g64, work = _oracle_grads(sd, x, y, torch.float64)
    _o32, _l32, g32, _w = _oracle_grads(sd, x, y, torch.float32)
    names = list(g64.keys())
    logits = net(x.to(dev))
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, y.to(dev))
    loss.backward()
    err = max_rel(logits, ref)
    print(f"DeepLabV3-CA train {shape}: logits max-rel {err:.3e} (oracle fp32 vs fp64 {max_rel(_o32, ref):.3e}), "
          f"loss {float(loss.detach()):.6f} vs {ref_loss:.6f}")
    assert err <= FWD_TOL
    assert abs(float(loss.detach()) - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    # BatchNorm buffers after the training-mode forward (the oracle updated `work` in place)
    got_sd = net.state_dict()
    for k in ("model.backbone.bn1.running_mean", "model.backbone.layer2.0.downsample.1.running_var",
              "model.backbone.layer4.2.bn2.running_var", "model.classifier.0.convs.4.2.running_mean",
              "model.classifier.0.convs.2.1.running_var", "model.classifier.2.running_mean"):
        assert max_rel(got_sd[k], work[k]) <= 1e-3, k
    assert int(got_sd["model.backbone.layer3.5.bn3.num_batches_tracked"]) == 1
    got = {k: p.grad for k, p in net.named_parameters()}
    hip_err, noise, bad = {}, {}, {}
    med_noise = float(np.median([rel_l2(g32[k], g64[k]) for k in names if float(g64[k].abs().max()) >= 1e-12]))
    for k in names:
        if float(g64[k].abs().max()) < 1e-12:
            assert float(got[k].abs().max()) < 1e-9, k
            continue
        hip_err[k], noise[k] = rel_l2(got[k], g64[k]), rel_l2(g32[k], g64[k])
        if hip_err[k] > max(3 * noise[k], 2 * med_noise):
            bad[k] = (hip_err[k], noise[k])
    print("largest HIP gradient rel-L2 vs fp64:", sorted(hip_err.items(), key=lambda kv: -kv[1])[:4])
    print(f"median rel-L2: HIP {np.median(list(hip_err.values())):.3e}, oracle fp32 {np.median(list(noise.values())):.3e}")
    assert not bad, bad
    assert np.median(list(hip_err.values())) <= 2 * np.median(list(noise.values())) + 1e-3
    assert hip_err["model.classifier.4.weight"] <= 1e-3 and hip_err["model.classifier.4.bias"] <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_relu_gate_in_the_gemm_epilogue(dev, dtype, monkeypatch):
    """The ReLU mask of a residual block's output applied by the GEMM that writes the block's incoming gradient
    (InsarIgemm.gate) against the pass of its own (insar_relu_gate_bwd): every parameter gradient bit for bit. With the
    BatchNorm-backward sums of the block's last unit taken in the same epilogue (over the gated values as stored) the sums
    are folded in another order: equal to rounding."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import deeplab, tape
    monkeypatch.setattr(tape, "MODE", "0")
    x, y = _input((2, 1, 64, 64), 11)
    grads, counts = [], []
    orig = deeplab.call
    for fuse, stats in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(deeplab, "GATE_FUSE", fuse)
        monkeypatch.setattr(deeplab, "GATE_STATS", stats)
        net, _sd = _make(dev, 33, dtype=dtype, p_drop=0.0)
        net.train()
        calls = []
        monkeypatch.setattr(deeplab, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
        loss = iu.CrossEntropyLoss(ignore_index=255)(net(x.to(dev)), y.to(dev))
        loss.backward()
        monkeypatch.setattr(deeplab, "call", orig)
        counts.append((calls.count("insar_relu_gate_bwd"), calls.count("insar_bnrelu_bwd_reduce")))
        grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})
    # fused: only layer1's last block (a strided consumer) gates in a pass of its own (layer4's last block: the broadcast that
    # writes its gradient last applies the mask)
    assert counts[0][0] == 16 and counts[1][0] == 1 and counts[2][0] == 1, counts
    assert counts[1][1] == counts[0][1] and counts[2][1] == counts[0][1] - 14, counts
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    worst = max((rel_l2(grads[2][k], grads[0][k]), k) for k in grads[0] if float(grads[0][k].abs().max()) > 1e-10)
    print("sums in the gating epilogue vs a pass of their own: worst gradient rel-L2", worst)
    assert worst[0] <= tol, worst


def test_deeplab_dropout_under_a_given_mask_and_adam_steps(dev):
    """Training mode with Dropout(0.5): the HIP path draws its own mask (torch's Philox stream cannot be reproduced); the
    oracle is given that mask. Then three Adam steps on both sides with externally supplied masks: the loss curves agree."""
    import insar_unet_ca_amd as iu
    net, sd = _make(dev, 31)
    net.train()
    shape = (4, 1, 64, 64)
    x, y = _input(shape, 17)
    logits = net(x.to(dev))
    plan = next(iter(net._plans.plans.values()))[0]
    mask = plan.drop_mask.permute(0, 3, 1, 2).contiguous().cpu()           # [B,256,h,w] of 0/1
    keep = float(mask.float().mean())
    assert 0.4 < keep < 0.6
    ref = dlo.forward(OrderedDict(sd), x, training=True, dropout_mask=mask)
    assert max_rel(logits, ref) <= FWD_TOL
    # Adam steps under given masks (p.grad / optimizer path of the drop-in)
    del logits
    net2, sd2 = _make(dev, 32)
    net2.train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net2.parameters(), lr=1e-4)
    names = [k for k in dlo.primary_keys(sd2) if dlo.is_param(k)]
    ref_params = [sd2[k].clone().requires_grad_(True) for k in names]
    ropt = torch.optim.Adam(ref_params, lr=1e-4)
    gen = torch.Generator().manual_seed(5)
    hip_losses, ref_losses = [], []
    plan2 = None
    for step in range(3):
        m = (torch.rand((4, 256, 8, 8), generator=gen) >= 0.5).to(torch.uint8)
        if plan2 is None:
            with torch.no_grad():
                net2(x.to(dev))                                              # builds the plan (and moves the BN buffers once)
            plan2 = next(iter(net2._plans.plans.values()))[0]
            plan2.external_mask = True
            net2.load_state_dict({k: v for k, v in sd2.items()})            # undo that forward's buffer update
        plan2.drop_mask.copy_(m.permute(0, 2, 3, 1).contiguous().to(dev))
        opt.zero_grad()
        l = crit(net2(x.to(dev)), y.to(dev))
        l.backward()
        opt.step()
        hip_losses.append(float(l))
        work = OrderedDict(sd2)
        for k, p in zip(names, ref_params):
            for kk in list(work.keys()):
                if work[kk] is sd2[k]:
                    work[kk] = p
        ropt.zero_grad()
        rl = dlo.cross_entropy(dlo.forward(work, x, training=True, dropout_mask=m), y)
        rl.backward()
        ropt.step()
        ref_losses.append(float(rl))
    print("DeepLabV3-CA 3 Adam steps: HIP", hip_losses, "oracle", ref_losses)
    assert all(abs(a - b) <= 2e-3 * max(1.0, abs(b)) for a, b in zip(hip_losses, ref_losses))


def test_deeplab_bf16_config5_geometry(dev):
    """BASELINE.json config 5: DeepLabV3-CA bf16, batch 16 of 1 x 256 x 256.
    Numerical contract: in EVAL mode (BatchNorm on running statistics) bf16 tracks the fp32 HIP path within the bf16
    gate. In TRAINING mode at random initialisation the 53-layer network is chaotic in ANY precision — the fp32 forward
    amplifies its own 6e-8 rounding to 1.3e-4 at the logits (2000x, test above, measured against float64), so bf16's
    4e-3 rounding saturates (measured 0.39 rel-L2 at the logits, growing x1.2 per bottleneck; tools/debug_deeplab_bf16.py)
    — so there the test checks what does hold: finite, bitwise reproducible, the loss at the fp32 value, and training."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(0, 16, 256, channels=1)
    x, y = x.to(dev), y.to(dev)
    crit = iu.CrossEntropyLoss(ignore_index=255)
    net32, _ = _make(dev, 41, p_drop=0.0)
    with torch.no_grad():
        ref_eval = net32.eval()(x).detach()
    ref_train_loss = float(crit(net32.train()(x), y))
    net32._plans.clear()
    del net32
    torch.cuda.empty_cache()
    net, _ = _make(dev, 41, dtype=torch.bfloat16, p_drop=0.0)
    with torch.no_grad():
        got_eval = net.eval()(x)
    err = max_rel(got_eval, ref_eval)
    agree = (got_eval.argmax(1) == ref_eval.argmax(1)).float().mean().item()
    print(f"config 5 bf16 vs fp32 HIP, eval mode: max-rel {err:.3e}, arg-max agreement {agree:.4f}")
    assert err <= 0.1 and agree >= 0.97
    net.train()
    opt = iu.Adam(net.parameters(), lr=1e-4)
    opt.zero_grad()
    l1 = crit(net(x), y)
    l1.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    opt.zero_grad()
    l2 = crit(net(x), y)
    l2.backward()
    print(f"config 5 training-mode loss: bf16 {float(l1):.5f}, fp32 {ref_train_loss:.5f}")
    assert float(l1) == float(l2) and all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    assert abs(float(l1) - ref_train_loss) <= 0.05 * ref_train_loss
    first = float(l2)
    for _ in range(6):
        opt.zero_grad()
        l = crit(net(x), y)
        l.backward()
        opt.step()
    assert float(l) < first


def _oracle_autocast_bf16(sd, x, y):
    """The oracle's training-mode forward + backward under torch's CPU autocast(bfloat16): what torch itself makes of this
    network in bf16 (convolutions in bf16, BatchNorm statistics in fp32), on the same weights and input."""
    names = [k for k in dlo.primary_keys(sd) if dlo.is_param(k)]
    work = OrderedDict((k, sd[k].clone()) for k in dlo.primary_keys(sd))
    leaves = []
    for k in names:
        work[k] = work[k].clone().requires_grad_(True)
        leaves.append(work[k])
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = dlo.forward(work, x, training=True)
    loss = dlo.cross_entropy(out.float(), y)
    grads = dict(zip(names, torch.autograd.grad(loss, leaves)))
    return out.detach().float(), float(loss), grads


@pytest.mark.parametrize("shape", [(4, 1, 64, 64), (2, 1, 128, 128)])
def test_deeplab_bf16_training_mode_against_torchs_own_bf16(dev, shape):
    """The numerical contract of config 5 in the arithmetic it is benchmarked in (bf16, TRAINING mode,
    DeepLabV3-ChannelAttention.py:140-162): reference = the oracle in float64; yardstick = torch's own bf16 (CPU autocast)
    on the same weights and input. 53 layers of batch statistics over small maps amplify any rounding, bf16's included, so
    both are far from the float64 result; the HIP bf16 path (bf16 storage, fp32 accumulation and statistics) must not be
    further from it than twice torch's own bf16: at the logits (rel-L2), in the loss, and per parameter-gradient tensor
    (rel-L2 within 2x torch's on that tensor, or within 2x torch's median: different roundings flip different ReLU / max-pool
    decisions), with the median over tensors within 2x torch's median."""
    import insar_unet_ca_amd as iu
    net, sd = _make(dev, 51, dtype=torch.bfloat16, p_drop=0.0)
    net.train()
    x, y = _input(shape, 23)
    ref, ref_loss, g64, _ = _oracle_grads(sd, x, y, torch.float64)
    t_out, t_loss, tg = _oracle_autocast_bf16(sd, x, y)
    logits = net(x.to(dev))
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, y.to(dev))
    loss.backward()
    e_hip, e_torch = rel_l2(logits, ref), rel_l2(t_out, ref)
    print(f"config-5 bf16 training mode {shape}: logits rel-L2 vs float64: HIP {e_hip:.3e}, torch autocast bf16 {e_torch:.3e}; "
          f"loss HIP {float(loss.detach()):.5f} torch-bf16 {t_loss:.5f} float64 {ref_loss:.5f}")
    assert e_hip <= 2 * e_torch + 1e-3
    assert abs(float(loss.detach()) - ref_loss) <= 2 * abs(t_loss - ref_loss) + 5e-3 * abs(ref_loss)
    got = {k: p.grad for k, p in net.named_parameters()}
    names = [k for k in g64 if float(g64[k].abs().max()) >= 1e-12]
    hip = {k: rel_l2(got[k], g64[k]) for k in names}
    tor = {k: rel_l2(tg[k], g64[k]) for k in names}
    med_h, med_t = float(np.median(list(hip.values()))), float(np.median(list(tor.values())))
    bad = {k: (hip[k], tor[k]) for k in names if hip[k] > max(2 * tor[k], 2 * med_t)}
    print(f"   gradient rel-L2 vs float64, median over {len(names)} tensors: HIP {med_h:.3e}, torch autocast bf16 {med_t:.3e}; "
          f"worst HIP {max(hip.values()):.3e}, worst torch {max(tor.values()):.3e}")
    assert med_h <= 2 * med_t
    assert len(bad) <= max(2, len(names) // 50), sorted(bad.items(), key=lambda kv: -kv[1][0])[:6]


@pytest.mark.parametrize("dtype,tol_curve,tol_final", [(torch.float32, 0.07, 0.02), (torch.bfloat16, 0.07, 0.05)])
def test_deeplab_convergence_tracks_the_oracle(dev, dtype, tol_curve, tol_final):
    """Thirty Adam(lr=1e-4) steps of DeepLabV3-CA on four 64 x 64 tiles, dropout off: the HIP loss curve (fp32 and bf16)
    against the oracle's own fp32 training run on the same weights and batches (torch CPU). After a few steps the two
    trajectories are different roundings of a chaotic system, so the gates are on the curve, not on its elements: every
    loss within tol_curve of the oracle's at the same step relative to the first loss, the mean of the last five within
    tol_final, and the loss must come down by at least as large a share as the oracle's does, less a fifth. Measured on
    MI355X: fp32 0.6568 -> 0.2678 (oracle 0.2708), largest difference along the curve 3.4 % of the first loss; bf16
    0.6509 -> 0.2842, 3.1 %: gates at twice the measured figures."""
    import insar_unet_ca_amd as iu
    net, sd = _make(dev, 61, dtype=dtype, p_drop=0.0)
    net.train()
    batches = [_input((4, 1, 64, 64), 30 + 4 * i) for i in range(3)]
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-4)
    names = [k for k in dlo.primary_keys(sd) if dlo.is_param(k)]
    ref_params = [sd[k].clone().requires_grad_(True) for k in names]
    ropt = torch.optim.Adam(ref_params, lr=1e-4)
    work = OrderedDict((k, sd[k].clone()) for k in dlo.primary_keys(sd))
    for k, p in zip(names, ref_params):
        work[k] = p
    hip, ref = [], []
    steps = 30
    for i in range(steps):
        x, y = batches[i % 3]
        opt.zero_grad()
        l = crit(net(x.to(dev)), y.to(dev))
        l.backward()
        opt.step()
        hip.append(float(l.detach()))
        ropt.zero_grad()
        rl = dlo.cross_entropy(dlo.forward(work, x, training=True), y)        # updates the BatchNorm buffers in `work`
        rl.backward()
        ropt.step()
        ref.append(float(rl))
    print(f"DeepLabV3-CA {dtype} convergence: HIP {hip[0]:.4f} -> {np.mean(hip[-5:]):.4f}, oracle {ref[0]:.4f} -> {np.mean(ref[-5:]):.4f}; "
          f"largest |difference| {max(abs(a - b) for a, b in zip(hip, ref)):.4f}")
    assert all(abs(a - b) <= tol_curve * ref[0] for a, b in zip(hip, ref)), list(zip(hip, ref))
    assert abs(np.mean(hip[-5:]) - np.mean(ref[-5:])) <= tol_final * ref[0]
    drop_ref, drop_hip = 1 - np.mean(ref[-5:]) / ref[0], 1 - np.mean(hip[-5:]) / hip[0]
    assert drop_ref > 0.02 and drop_hip >= 0.8 * drop_ref
