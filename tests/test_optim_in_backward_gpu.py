"""GPU: optim.Adam.fuse_into_backward — the optimizer update and the weight re-layout of a backward stage issued on the side
stream as soon as that stage's gradients are enqueued — against the reference's order (loss.backward() then optimizer.step(),
Unet-ChannalAttention.py:345-346): the same arithmetic per parameter, so losses, parameters, BatchNorm buffers and optimizer
state must be bit for bit those of the plain step."""
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


def _run(dev, model_name, dtype, fused, steps, batches):
    import insar_unet_ca_amd as iu
    torch.manual_seed(9)
    if model_name == "unet":
        net = iu.UNet(2, 2, True, compute_dtype=dtype)
        crit = iu.DiceCELoss(ignore_index=255)
    else:
        net = iu.DeepLabV3_SingleChannel_Attn(2, "resnet50", False, compute_dtype=dtype)
        net.aspp.project[3].p = 0.0
        crit = iu.CrossEntropyLoss(ignore_index=255)
    net = net.to(dev).train()
    opt = iu.Adam(net.parameters(), lr=1e-3)
    if fused:
        opt.fuse_into_backward(net)
    losses, early = [], []
    for i in range(steps):
        x, y = batches[i % len(batches)]
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), y)
        loss.backward()
        early.append(opt._early_stages if fused else 0)
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    # one more forward in eval mode: it must see the updated weights through the re-laid-out copies
    with torch.no_grad():
        probe = net.eval()(batches[0][0]).clone()
    return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, opt.state_dict(), probe, early


@pytest.mark.parametrize("model_name,dtype,size,chan", [("unet", torch.bfloat16, 64, 2), ("unet", torch.float32, 32, 2),
                                                      ("deeplab", torch.bfloat16, 64, 1)])
def test_update_inside_backward_is_bitwise_the_plain_step(dev, model_name, dtype, size, chan):
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, size, channels=chan)) for i in range(3)]
    steps = 8
    lp, sdp, osp, pp, _ = _run(dev, model_name, dtype, False, steps, batches)
    lf, sdf, osf, pf, early = _run(dev, model_name, dtype, True, steps, batches)
    assert lp == lf, (lp, lf)
    for k in sdp:
        assert torch.equal(sdp[k], sdf[k]), k
    assert torch.equal(pp, pf)
    for i in osp["state"]:
        assert float(osp["state"][i]["step"]) == float(osf["state"][i]["step"]) == steps
        assert torch.equal(osp["state"][i]["exp_avg"], osf["state"][i]["exp_avg"])
        assert torch.equal(osp["state"][i]["exp_avg_sq"], osf["state"][i]["exp_avg_sq"])
    # the first steps build the optimizer state through the plain path; from then on every stage updates inside backward
    assert early[0] == 0 and early[-1] >= 6, early


def test_second_backward_before_step_is_refused(dev):
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    x, y = (t.to(dev) for t in make_batch(0, 2, 32))
    torch.manual_seed(1)
    net = iu.UNet(2, 2, True).to(dev).train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-3).fuse_into_backward(net)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        crit(net(x), y).backward()
        opt.step()
    crit(net(x), y).backward()
    with pytest.raises(iu.InsarError, match="second backward"):
        crit(net(x), y).backward()
