"""Pin the CPU oracle (oracle/unet_ca_oracle.py) against golden vectors produced by the
reference itself (oracle/gen_golden.py -> tests/golden/*.npz). CPU only."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import closed_form as cf
from oracle import unet_ca_oracle as orc
from tests.helpers import check_summary

TOL = 2e-5


def _filled(template):
    return cf.fill_state_dict(template)


def _leafify(sd):
    work, leaves = OrderedDict(sd), {}
    for k in sd:
        if orc.is_param(k):
            work[k] = sd[k].detach().clone().requires_grad_(True)
            leaves[k] = work[k]
    return work, leaves


def test_state_dict_contract(golden):
    g3 = golden("g3_unet")
    sd = orc.state_dict_template(2, 2, True)
    assert list(sd.keys()) == [str(k) for k in g3["state_dict_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g3["state_dict_shapes"]]
    assert len(sd) == 154
    assert sum(v.numel() for k, v in sd.items() if orc.is_param(k)) == 31261122


@pytest.mark.parametrize("tag,c,shape,salt", [("se64", 64, (2, 64, 8, 8), 0.0), ("se128", 128, (3, 128, 4, 4), 0.3)])
def test_se_layer(golden, tag, c, shape, salt):
    g1 = golden("g1_blocks")
    tmpl = OrderedDict([("fc.0.weight", torch.zeros(c // 16, c)), ("fc.2.weight", torch.zeros(c, c // 16))])
    sd = _filled(tmpl)
    w1 = sd["fc.0.weight"].requires_grad_(True)
    w2 = sd["fc.2.weight"].requires_grad_(True)
    x = cf.make_input(shape, salt).requires_grad_(True)
    out = orc.se_layer(x, w1, w2)
    out.backward(cf.make_grad(out.shape))
    check_summary(g1, f"{tag}/step0/out", out, TOL)
    check_summary(g1, f"{tag}/step0/dx", x.grad, TOL)
    check_summary(g1, f"{tag}/step0/grad/fc.0.weight", w1.grad, TOL)
    check_summary(g1, f"{tag}/step0/grad/fc.2.weight", w2.grad, TOL)


@pytest.mark.parametrize("tag,c,shape", [("cam256", 256, (2, 256, 8, 8)), ("cam64_ties", 64, (3, 64, 12, 20))])
def test_channel_attention_module(golden, tag, c, shape):
    """G7: ChannelAttentionModule of the DeepLabV3 variant (config 5), incl. exact ties for the maximum."""
    g7 = golden("g7_cam")
    tmpl = OrderedDict([("mlp.0.weight", torch.zeros(c // 16, c, 1, 1)), ("mlp.2.weight", torch.zeros(c, c // 16, 1, 1))])
    sd = _filled(tmpl)
    w1 = sd["mlp.0.weight"].requires_grad_(True)
    w2 = sd["mlp.2.weight"].requires_grad_(True)
    x = torch.from_numpy(g7[f"{tag}/x"]) if f"{tag}/x" in g7.files else cf.make_input(shape, 0.4)
    x = x.clone().requires_grad_(True)
    out = orc.cam_layer(x, w1, w2)
    out.backward(cf.make_grad(out.shape))
    check_summary(g7, f"{tag}/step0/out", out, TOL)
    check_summary(g7, f"{tag}/step0/dx", x.grad, TOL)
    check_summary(g7, f"{tag}/step0/grad/mlp.0.weight", w1.grad, TOL)
    check_summary(g7, f"{tag}/step0/grad/mlp.2.weight", w2.grad, TOL)


def _dc_template(cin, cout, use_se):
    return OrderedDict((k[len("blk."):], torch.zeros(s, dtype=torch.int64 if k.endswith("tracked") else torch.float32))
                       for k, s in orc._double_conv_entries("blk", cin, cout, use_se))


@pytest.mark.parametrize("tag,cin,cout,use_se,shape,salt,training,steps,need_dx", [
    ("dc_2_64_se_train", 2, 64, True, (2, 2, 16, 16), 0.0, True, 2, False),
    ("dc_64_128_se_train", 64, 128, True, (2, 64, 16, 16), 0.0, True, 2, True),
    ("dc_64_128_se_eval", 64, 128, True, (2, 64, 16, 16), 0.0, False, 1, True),
    ("dc_128_64_plain_train", 128, 64, False, (2, 128, 16, 16), 0.0, True, 1, True),
    ("dc_128_64_se_ragged", 128, 64, True, (1, 128, 8, 24), 0.7, True, 1, True),
])
def test_double_conv(golden, tag, cin, cout, use_se, shape, salt, training, steps, need_dx):
    g1 = golden("g1_blocks")
    sd = _filled(_dc_template(cin, cout, use_se))
    sd = OrderedDict(("blk." + k, v) for k, v in sd.items())
    for s in range(steps):
        work, leaves = _leafify(sd)
        x = cf.make_input(shape, salt).requires_grad_(need_dx)
        out = orc.double_conv(x, work, "blk", use_se, training)
        out.backward(cf.make_grad(out.shape))
        pre = f"{tag}/step{s}"
        check_summary(g1, f"{pre}/out", out, TOL)
        if need_dx:
            check_summary(g1, f"{pre}/dx", x.grad, TOL)
        for k, leaf in leaves.items():
            check_summary(g1, f"{pre}/grad/{k[4:]}", leaf.grad, 1e-4)
        for k in sd:
            if not orc.is_param(k):
                sd[k] = work[k]
                check_summary(g1, f"{pre}/buf/{k[4:]}", work[k], TOL)


def test_pool_and_convT(golden):
    g2 = golden("g2_resample")
    x = torch.from_numpy(g2["pool/x"]).requires_grad_(True)
    out = F.max_pool2d(x, 2)
    out.backward(cf.make_grad(out.shape))
    check_summary(g2, "pool/out", out, 1e-7)
    check_summary(g2, "pool/dx", x.grad, 1e-7)
    tmpl = OrderedDict([("weight", torch.zeros(128, 64, 2, 2)), ("bias", torch.zeros(64))])
    sd = _filled(tmpl)
    w, b = sd["weight"].requires_grad_(True), sd["bias"].requires_grad_(True)
    xi = cf.make_input((2, 128, 8, 8)).requires_grad_(True)
    out = F.conv_transpose2d(xi, w, b, stride=2)
    out.backward(cf.make_grad(out.shape))
    check_summary(g2, "convT_128_64/step0/out", out, TOL)
    check_summary(g2, "convT_128_64/step0/dx", xi.grad, TOL)
    check_summary(g2, "convT_128_64/step0/grad/weight", w.grad, TOL)
    check_summary(g2, "convT_128_64/step0/grad/bias", b.grad, TOL)


@pytest.mark.parametrize("tag,shape,training", [
    ("b2_64_train", (2, 2, 64, 64), True),
    ("b3_48x80_train", (3, 2, 48, 80), True),
    ("b1_256_eval", (1, 2, 256, 256), False),
    ("b1_256_train", (1, 2, 256, 256), True),
])
def test_unet(golden, tag, shape, training):
    g3 = golden("g3_unet")
    sd = _filled(orc.state_dict_template(2, 2, True))
    work, leaves = _leafify(sd)
    x = cf.make_input(shape)
    tgt = cf.make_target((shape[0], shape[2], shape[3]), ignore_every=13)
    with torch.set_grad_enabled(training):
        logits = orc.unet_forward(work, x, use_se=True, training=training)
        loss = orc.cross_entropy(logits, tgt)
    check_summary(g3, f"{tag}/logits", logits, TOL)
    assert abs(float(loss.detach()) - float(g3[f"{tag}/loss"])) < 1e-5
    m = orc.compute_metrics(logits.detach(), tgt, 2)
    np.testing.assert_allclose([m[k] for k in ("acc", "miou", "mpa", "mf1")], g3[f"{tag}/metrics"], atol=1e-12)
    if training:
        loss.backward()
        for k, leaf in leaves.items():
            check_summary(g3, f"{tag}/grad/{k}", leaf.grad, 2e-4)
        for k in sd:
            if not orc.is_param(k) and not k.endswith("num_batches_tracked"):
                check_summary(g3, f"{tag}/buf/{k}", work[k], TOL)


def test_unet_without_se(golden):
    g3 = golden("g3_unet")
    sd = _filled(orc.state_dict_template(2, 2, False))
    work, leaves = _leafify(sd)
    logits = orc.unet_forward(work, cf.make_input((2, 2, 32, 32)), use_se=False, training=True)
    loss = orc.cross_entropy(logits, cf.make_target((2, 32, 32)))
    loss.backward()
    check_summary(g3, "nose_b2_32_train/logits", logits, TOL)
    assert abs(float(loss.detach()) - float(g3["nose_b2_32_train/loss"])) < 1e-5
    for k, leaf in leaves.items():
        check_summary(g3, f"nose_b2_32_train/grad/{k}", leaf.grad, 2e-4)


def test_adam_five_steps(golden):
    g4 = golden("g4_adam")
    sd = _filled(orc.state_dict_template(2, 2, True))
    start = {k: v.clone() for k, v in sd.items()}
    state, losses = {}, []
    for step in range(5):
        x = cf.make_input((2, 2, 64, 64), salt=0.37 * step)
        tgt = cf.make_target((2, 64, 64))
        loss, _ = orc.train_step(sd, state, x, tgt, use_se=True, lr=1e-4)
        losses.append(loss)
    np.testing.assert_allclose(losses, g4["losses"], rtol=2e-5)
    for k, v in sd.items():
        if v.dtype == torch.float32:
            check_summary(g4, f"final/{k}", v, 1e-5)
            # Adam's per-step delta is +-lr*O(1): compare deltas against lr scale
            check_summary(g4, f"delta/{k}", v - start[k], 5e-2)


def test_cross_entropy(golden):
    g5 = golden("g5_ce")
    lg = (cf.make_input((2, 2, 16, 16), 0.9) * 3.0).requires_grad_(True)
    tgt = cf.make_target((2, 16, 16), ignore_every=5)
    loss = orc.cross_entropy(lg, tgt)
    loss.backward()
    assert abs(float(loss) - float(g5["loss"])) < 1e-6
    np.testing.assert_allclose(lg.grad.numpy(), g5["dlogits"], atol=1e-8)
    lg3 = (cf.make_input((2, 3, 8, 8), 0.1) * 2.0).requires_grad_(True)
    tgt3 = torch.from_numpy(g5["target3"])
    loss3 = orc.cross_entropy(lg3, tgt3)
    loss3.backward()
    assert abs(float(loss3) - float(g5["loss3"])) < 1e-6
    np.testing.assert_allclose(lg3.grad.numpy(), g5["dlogits3"], atol=1e-8)


@pytest.mark.parametrize("case", ["three_of_four", "all_tie", "class1_absent", "ignore255"])
def test_metrics_kat(golden, case):
    g6 = golden("g6_metrics")
    m = orc.compute_metrics(torch.from_numpy(g6[f"{case}/logits"]), torch.from_numpy(g6[f"{case}/mask"]), 2)
    np.testing.assert_allclose([m[k] for k in ("acc", "miou", "mpa", "mf1")], g6[f"{case}/expect"], atol=1e-12)


def test_metrics_kat_values(golden):
    """The survey's hand-derived values for the quirky acc formula."""
    g6 = golden("g6_metrics")
    np.testing.assert_allclose(g6["three_of_four/expect"], [0.6, 7 / 12, 0.75, 11 / 15], atol=1e-12)
    np.testing.assert_allclose(g6["all_tie/expect"], [1 / 3, 0.25, 0.5, 1 / 3], atol=1e-12)


def test_dp_mean_of_shard_grads(golden):
    g9 = golden("g9_dp")
    acc = None
    for r in range(2):
        sd = _filled(orc.state_dict_template(2, 2, True))
        work, leaves = _leafify(sd)
        x = cf.make_input((2, 2, 32, 32), salt=1.1 * r)
        loss = orc.cross_entropy(orc.unet_forward(work, x, True, True), cf.make_target((2, 32, 32)))
        loss.backward()
        if acc is None:
            acc = {k: 0.5 * v.grad for k, v in leaves.items()}
        else:
            for k, v in leaves.items():
                acc[k] += 0.5 * v.grad
    for k, v in acc.items():
        check_summary(g9, f"mean_grad/{k}", v, 2e-4)


def test_unet_random_fixture(golden):
    """G3r: generic-position (PCG64) weights/inputs, the fixture the gradient parity of the HIP path is pinned on."""
    g = golden("g3r_unet_random")
    sd = cf.fill_state_dict_random(orc.state_dict_template(2, 2, True), seed=7)
    work, leaves = _leafify(sd)
    x = cf.make_input_random((2, 2, 64, 64), seed=11)
    tgt = cf.make_target_random((2, 64, 64), seed=13, ignore_frac=0.05)
    logits = orc.unet_forward(work, x, use_se=True, training=True)
    loss = orc.cross_entropy(logits, tgt)
    loss.backward()
    check_summary(g, "b2_64_train/logits", logits, TOL)
    assert abs(float(loss.detach()) - float(g["b2_64_train/loss"])) < 1e-5
    for k, leaf in leaves.items():
        check_summary(g, f"b2_64_train/grad/{k}", leaf.grad, 2e-4)
    sd = cf.fill_state_dict_random(orc.state_dict_template(2, 2, True), seed=7)
    state, losses = {}, []
    for step in range(5):
        xs = cf.make_input_random((2, 2, 64, 64), seed=100 + step)
        ts = cf.make_target_random((2, 64, 64), seed=200 + step)
        l, _ = orc.train_step(sd, state, xs, ts, use_se=True, lr=1e-4)
        losses.append(l)
    np.testing.assert_allclose(losses, g["adam/losses"], rtol=2e-5)
