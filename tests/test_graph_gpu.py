"""GPU: the training step replayed from a captured hipGraph (insar_unet_ca_amd.GraphedTrainStep) is the eager step:
same kernels in the same order on the same two streams, Adam's step count and bias corrections on the device."""
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


def _run(dev, model_name, dtype, graphed, steps, batches):
    import insar_unet_ca_amd as iu
    torch.manual_seed(3)
    if model_name == "unet":
        net = iu.UNet(2, 2, True, compute_dtype=dtype)
        crit = iu.DiceCELoss(ignore_index=255)
    else:
        net = iu.DeepLabV3_SingleChannel_Attn(2, "resnet50", False, compute_dtype=dtype)
        net.aspp.project[3].p = 0.0                 # the dropout mask depends on a per-forward counter: compare without it
        crit = iu.CrossEntropyLoss(ignore_index=255)
    net = net.to(dev).train()
    opt = iu.Adam(net.parameters(), lr=1e-3)
    losses = []
    if graphed:
        step = iu.GraphedTrainStep(net, crit, opt, batches[0][0], batches[0][1], warmup=2)
        done = step.warmup_steps
        for i in range(done, steps):
            x, y = batches[i % len(batches)]
            losses.append(float(step(x, y)))
    else:
        for i in range(steps):
            x, y = batches[0] if i < 2 else batches[i % len(batches)]      # the warm-up of the graphed run uses batch 0
            opt.zero_grad(set_to_none=True)
            l = crit(net(x), y)
            l.backward()
            opt.step()
            if i >= 2:
                losses.append(float(l))
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    osd = opt.state_dict()
    return losses, sd, osd, {k: p.grad.clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize("model_name,dtype,size,chan", [("unet", torch.bfloat16, 64, 2), ("unet", torch.float32, 32, 2),
                                                      ("deeplab", torch.bfloat16, 64, 1)])
def test_graph_replay_is_bitwise_the_eager_step(dev, model_name, dtype, size, chan):
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, size, channels=chan)) for i in range(3)]
    steps = 7
    le, sde, osde, ge = _run(dev, model_name, dtype, False, steps, batches)
    lg, sdg, osdg, gg = _run(dev, model_name, dtype, True, steps, batches)
    assert le == lg, (le, lg)
    for k in sde:
        assert torch.equal(sde[k], sdg[k]), k                      # parameters AND BatchNorm buffers (num_batches_tracked too)
    for k in ge:
        assert torch.equal(ge[k], gg[k]), k                        # .grad holds the last step's gradient
    # the optimizer state dict still counts the steps per parameter, like torch.optim.Adam's
    st_e, st_g = osde["state"], osdg["state"]
    assert len(st_e) == len(st_g) > 0
    for i in st_e:
        assert float(st_e[i]["step"]) == float(st_g[i]["step"]) == steps
        assert torch.equal(st_e[i]["exp_avg"], st_g[i]["exp_avg"]) and torch.equal(st_e[i]["exp_avg_sq"], st_g[i]["exp_avg_sq"])


def test_graph_refuses_what_it_cannot_capture(dev):
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    net = iu.UNet(2, 2, True).to(dev).eval()
    x, y = (t.to(dev) for t in make_batch(0, 2, 32))
    with pytest.raises(iu.InsarError, match="TRAINING"):
        iu.GraphedTrainStep(net, iu.CrossEntropyLoss(ignore_index=255), iu.Adam(net.parameters()), x, y)
    with pytest.raises(iu.InsarError, match="no CPU fallback"):
        iu.GraphedTrainStep(net.train(), iu.CrossEntropyLoss(ignore_index=255), iu.Adam(net.parameters()), x.cpu(), y.cpu())


def test_optimizer_state_loaded_after_the_capture_is_what_the_replay_uses(dev):
    """Resume scenario: a GraphedTrainStep exists, then model and optimizer state are loaded from a checkpoint.
    Adam.load_state_dict restores into the tensors the captured kernels point at (moments, device-side step state), so the
    replayed steps continue the CHECKPOINT's trajectory bit for bit; a load that has to re-allocate state makes the next
    replay raise instead of updating freed memory."""
    import copy
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, 32, channels=2)) for i in range(3)]

    def fresh():
        torch.manual_seed(5)
        net = iu.UNet(2, 2, True, compute_dtype=torch.float32).to(dev).train()
        return net, iu.DiceCELoss(ignore_index=255), iu.Adam(net.parameters(), lr=1e-3)

    # run A: 5 eager steps, checkpoint, 3 more eager steps
    net, crit, opt = fresh()
    def eager(n, first):
        out = []
        for i in range(first, first + n):
            x, y = batches[i % 3]
            opt.zero_grad(set_to_none=True)
            l = crit(net(x), y); l.backward(); opt.step(); out.append(float(l))
        return out
    eager(5, 0)
    ck_model = copy.deepcopy(net.state_dict()); ck_opt = copy.deepcopy(opt.state_dict())
    tail_a = eager(3, 5)
    params_a = {k: v.clone() for k, v in net.state_dict().items()}
    # run B: another trajectory, captured; then the checkpoint is loaded into the live objects
    net, crit, opt = fresh()
    step = iu.GraphedTrainStep(net, crit, opt, batches[2][0], batches[2][1], warmup=2)
    step(*batches[1])
    net.load_state_dict(ck_model)
    gen = opt.generation
    opt.load_state_dict(ck_opt)
    assert opt.generation == gen, "same-shape state must be restored in place"
    tail_b = [float(step(*batches[i % 3])) for i in range(5, 8)]
    assert tail_a == tail_b, (tail_a, tail_b)
    for k, v in net.state_dict().items():
        assert torch.equal(v, params_a[k]), k
    assert float(opt.state_dict()["state"][0]["step"]) == 8
    # a load that re-allocates (fresh optimizer state for other shapes) is refused at replay time
    opt.generation += 1
    with pytest.raises(iu.InsarError, match="re-allocated"):
        step(*batches[0])
