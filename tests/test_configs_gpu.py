"""GPU parity at the FULL sizes of BASELINE.json's configurations (run with -m gpu on an MI355X):

  config 2: U-Net-CA, batch 16 x 2 x 256 x 256 — fp32 logits and loss against the CPU oracle (<= 1e-3, north_star),
            bf16 (the benchmarked arithmetic) against the same oracle under the calibrated bf16 gate;
  config 4: U-Net-CA fp32, 512 x 512 tiles — a batch-2 run against the oracle (<= 1e-3) and the batch-8 training
            step (forward + CE + backward + Adam) through size-independent properties.

The oracle (oracle/unet_ca_oracle.py, pinned to the reference by tests/golden) runs on the box's host cores: one
training-mode forward of 16 tiles of 256 x 256 takes a few seconds.
"""
from collections import OrderedDict

import pytest
import torch

from oracle import unet_ca_oracle as orc
from tests.helpers import max_rel

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-3          # north_star: fp32 forward <= 1e-3 rel
# bf16 gate at torch-default initialisation (the benchmark's): SURVEY 8d probed torch's own bf16 at 2.2e-2 max-rel and
# 99.4 % arg-max agreement on such weights; the gate leaves the same 2x head-room the G3r gate (8.5e-2) has over its
# measured 5.8e-2.
BF16_TOL, BF16_ARGMAX = 5e-2, 0.985


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from insar_unet_ca_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _net_and_state(dev, dtype, seed=0):
    import insar_unet_ca_amd as iu
    torch.manual_seed(seed)
    net = iu.UNet(2, 2, True, compute_dtype=dtype)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    return net.to(dev).train(), sd


def _release(net):
    net._plans.clear()
    torch.cuda.empty_cache()


def test_config2_full_size_fp32_and_bf16_against_oracle(dev):
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    x, y = make_batch(0, 16, 256)
    net, sd = _net_and_state(dev, torch.float32)
    with torch.no_grad():
        ref = orc.unet_forward(OrderedDict((k, v.clone()) for k, v in sd.items()), x, use_se=True, training=True)
        ref_ce = float(orc.cross_entropy(ref, y))
    logits = net(x.to(dev))
    loss = iu.CrossEntropyLoss(ignore_index=255)(logits, y.to(dev))
    err = max_rel(logits, ref)
    print(f"config 2 fp32 16x2x256x256: logits max-rel {err:.3e}, CE {float(loss):.6f} vs oracle {ref_ce:.6f}")
    assert err <= FWD_TOL
    assert abs(float(loss) - ref_ce) <= 1e-4 * max(1.0, abs(ref_ce))
    # BatchNorm running statistics after this one training-mode forward (momentum 0.1, unbiased variance)
    rm = net.state_dict()["down4.1.double_conv.4.running_mean"].cpu()
    with torch.no_grad():
        sd2 = OrderedDict((k, v.clone()) for k, v in sd.items())
        orc.unet_forward(sd2, x, use_se=True, training=True)
    assert max_rel(rm, sd2["down4.1.double_conv.4.running_mean"]) <= 1e-3
    _release(net)
    del net

    net16, _ = _net_and_state(dev, torch.bfloat16)          # same seed: same weights
    l16 = net16(x.to(dev))
    err16 = max_rel(l16, ref)
    agree = (l16.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
    loss16 = float(iu.DiceCELoss(ignore_index=255)(l16, y.to(dev)))
    print(f"config 2 bf16 16x2x256x256: logits max-rel {err16:.3e}, arg-max agreement {agree:.4f}")
    assert err16 <= BF16_TOL and agree >= BF16_ARGMAX
    assert loss16 == loss16
    _release(net16)


def test_config4_fp32_512_against_oracle_and_properties(dev):
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    # (a) batch 2 of 512 x 512 tiles against the oracle
    net, sd = _net_and_state(dev, torch.float32, seed=3)
    x2, y2 = make_batch(100, 2, 512)
    with torch.no_grad():
        ref = orc.unet_forward(OrderedDict((k, v.clone()) for k, v in sd.items()), x2, use_se=True, training=True)
        ref_ce = float(orc.cross_entropy(ref, y2))
    crit = iu.CrossEntropyLoss(ignore_index=255)
    logits = net(x2.to(dev))
    loss = crit(logits, y2.to(dev))
    err = max_rel(logits, ref)
    print(f"config 4 fp32 2x2x512x512: logits max-rel {err:.3e}, CE {float(loss):.6f} vs oracle {ref_ce:.6f}")
    assert err <= FWD_TOL
    assert abs(float(loss) - ref_ce) <= 1e-4 * max(1.0, abs(ref_ce))
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    net._plans.clear()
    # (b) the configuration itself: batch 8, forward + CE + backward + Adam
    x, y = make_batch(0, 8, 512)
    x, y = x.to(dev), y.to(dev)
    opt = iu.Adam(net.parameters(), lr=1e-4)

    def one_step():
        opt.zero_grad()
        out = net(x)
        l = crit(out, y)
        l.backward()
        return out, l

    l1, loss1 = one_step()
    g1 = [p.grad.clone() for p in net.parameters()]
    l2, loss2 = one_step()
    assert torch.equal(l1, l2) and float(loss1) == float(loss2)                 # no atomics: bitwise reproducible
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))
    plan = net._plan(x)
    assert plan.B == 8 and plan.H == 512 and plan.ctx.dtype == torch.float32
    u = plan.enc[0].u1                                                           # BN output of the first unit normalised
    zpre = u.y.nchw() * u.scale.view(1, -1, 1, 1) + u.shift.view(1, -1, 1, 1)
    assert float(zpre.mean((0, 2, 3)).abs().max()) < 1e-3
    assert float((zpre.var((0, 2, 3), unbiased=False) - 1).abs().max()) < 1e-3
    assert float(net.inc.double_conv[0].bias.grad.abs().max()) == 0.0          # pre-BN conv bias: exactly zero
    # the mean of the CE gradient over classes is 0 => outc.bias gradients cancel
    assert abs(float(net.outc.bias.grad.sum())) <= 1e-5
    first = float(loss2)
    for _ in range(6):
        _, l = one_step()
        opt.step()
    assert float(l) < first
    _release(net)
