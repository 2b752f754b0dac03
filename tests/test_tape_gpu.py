"""GPU: launch tapes (insar_unet_ca_amd/tape.py) — the training-mode forward / backward of a plan replayed from a recorded
launch list — against the ordinary launch code: losses, parameters, BatchNorm buffers, gradients and optimizer state bit for bit,
through validation passes in between (eval mode is never taped), a second batch geometry, dropout, and the verify mode that
re-records every 16th call and compares it with the tape."""
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


def _run(dev, model_name, dtype, mode, steps, batches, monkeypatch, p_drop=0.0, accum=1, set_to_none=True, reassign_at=None):
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import tape
    monkeypatch.setattr(tape, "MODE", mode)
    torch.manual_seed(4)
    if model_name == "unet":
        net = iu.UNet(2, 2, True, compute_dtype=dtype)
        crit = iu.DiceCELoss(ignore_index=255)
    else:
        net = iu.DeepLabV3_SingleChannel_Attn(2, "resnet50", False, compute_dtype=dtype)
        net.aspp.project[3].p = p_drop
        crit = iu.CrossEntropyLoss(ignore_index=255)
    net = net.to(dev).train()
    opt = iu.Adam(net.parameters(), lr=1e-3)
    losses, evals = [], []
    for i in range(steps):
        if reassign_at is not None and i == reassign_at:       # new storage under every parameter and BatchNorm buffer
            for prm in net.parameters():
                prm.data = prm.data.clone()
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean = m.running_mean.clone()
                    m.running_var = m.running_var.clone()
        opt.zero_grad(set_to_none=set_to_none)
        for k in range(accum):                                  # gradient accumulation: several backwards per optimizer step
            x, y = batches[(i * accum + k) % len(batches)]
            loss = crit(net(x), y)
            loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if i % 5 == 4:                                      # a validation pass between training steps
            net.eval()
            with torch.no_grad():
                evals.append(net(batches[0][0]).clone())
            net.train()
    torch.cuda.synchronize()
    plans = [pl for lst in net._plans.plans.values() for pl in (lst if isinstance(lst, list) else [lst])]
    reports = [pl.tape_report() for pl in plans if hasattr(pl, "tape_report")]
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, grads, opt.state_dict(), evals, reports


def _assert_equal(a, b):
    la, sda, ga, osa, ea, _ = a
    lb, sdb, gb, osb, eb, _ = b
    assert la == lb, (la, lb)
    for k in sda:
        assert torch.equal(sda[k], sdb[k]), k
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    for x, y in zip(ea, eb):
        assert torch.equal(x, y)
    for i in osa["state"]:
        assert float(osa["state"][i]["step"]) == float(osb["state"][i]["step"])
        assert torch.equal(osa["state"][i]["exp_avg"], osb["state"][i]["exp_avg"])
        assert torch.equal(osa["state"][i]["exp_avg_sq"], osb["state"][i]["exp_avg_sq"])


@pytest.mark.parametrize("model_name,dtype,size,chan,p_drop", [("unet", torch.bfloat16, 64, 2, 0.0), ("unet", torch.float32, 32, 2, 0.0),
                                                             ("deeplab", torch.bfloat16, 64, 1, 0.0)])
def test_taped_training_is_bitwise_the_ordinary_launch_code(dev, model_name, dtype, size, chan, p_drop, monkeypatch):
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, size, channels=chan)) for i in range(3)]
    steps = 12
    off = _run(dev, model_name, dtype, "0", steps, batches, monkeypatch, p_drop)
    on = _run(dev, model_name, dtype, "1", steps, batches, monkeypatch, p_drop)
    _assert_equal(off, on)
    states = [v for rep in on[5] for v in rep.values()]
    assert states and all(s.startswith("replaying") for s in states), on[5]       # forward and backward tapes both took over
    assert not any(off[5])                                                         # and none without the switch


def test_two_geometries_and_verify_mode(dev, monkeypatch):
    """Two batch geometries (two plans, two pairs of tapes) alternating, with INSAR_TAPE=verify: every 16th call of a tape is
    re-recorded from the live code and must equal the tape launch for launch (it raises otherwise)."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import tape
    from insar_unet_ca_amd.data import make_batch
    monkeypatch.setattr(tape, "MODE", "verify")
    a = tuple(t.to(dev) for t in make_batch(0, 4, 32))
    b = tuple(t.to(dev) for t in make_batch(8, 2, 64))
    torch.manual_seed(2)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-3)
    for i in range(40):
        x, y = a if i % 2 == 0 else b
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), y)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    plans = [pl for lst in net._plans.plans.values() for pl in (lst if isinstance(lst, list) else [lst])]
    assert len(plans) == 2
    for pl in plans:
        rep = pl.tape_report()
        assert len(rep) == 2 and all(v.startswith("replaying") for v in rep.values()), rep


def test_dropout_advances_under_the_tape(dev, monkeypatch):
    """DeepLabV3-CA with Dropout(0.5): the mask is keyed by a device-side counter that the taped forward advances like the
    ordinary one — masks differ from step to step and the trajectory equals the untaped one."""
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, 64, channels=1)) for i in range(2)]
    off = _run(dev, "deeplab", torch.bfloat16, "0", 8, batches, monkeypatch, 0.5)
    on = _run(dev, "deeplab", torch.bfloat16, "1", 8, batches, monkeypatch, 0.5)
    assert off[0] == on[0], (off[0], on[0])
    for k in off[1]:
        assert torch.equal(off[1][k], on[1][k]), k


@pytest.mark.parametrize("accum,set_to_none", [(4, True), (3, False), (1, False)])
def test_tape_with_gradient_accumulation(dev, accum, set_to_none, monkeypatch):
    """Several training-mode forwards / backwards before the FIRST optimizer step: both recordings are made while no weight is
    stale, so a tape that baked "no re-layout" into itself would train on stale bf16 copies after the first step (ADVICE r3,
    high). The re-layout check is a live op of the tape (tape.tape_live): trajectory bit for bit the untaped one."""
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, 32)) for i in range(3)]
    off = _run(dev, "unet", torch.bfloat16, "0", 6, batches, monkeypatch, accum=accum, set_to_none=set_to_none)
    on = _run(dev, "unet", torch.bfloat16, "1", 6, batches, monkeypatch, accum=accum, set_to_none=set_to_none)
    _assert_equal(off, on)
    # the forward tape and a backward tape took over (with live .grad views backward alternates between the two gradient
    # buffers or stays on one of them: the tape of the other may never leave its recording state)
    rep = {k: v for r in on[5] for k, v in r.items()}
    assert all(v.startswith("replaying") for k, v in rep.items() if k[0] == "f") and any(k[0] == "f" for k in rep), rep
    assert any(v.startswith("replaying") for k, v in rep.items() if k[0] == "b"), rep
    assert off[0][0] != off[0][-1]                           # and the weights did move


@pytest.mark.parametrize("model_name,chan", [("unet", 2), ("deeplab", 1)])
def test_tape_is_dropped_when_parameter_storage_moves(dev, model_name, chan, monkeypatch):
    """p.data = ..., a swapped running_mean: the ordinary code re-reads every pointer per call; a tape holds them in prebuilt
    arguments. The tape's storage fingerprint notices, the tape is recorded again."""
    from insar_unet_ca_amd.data import make_batch
    batches = [tuple(t.to(dev) for t in make_batch(4 * i, 4, 64, channels=chan)) for i in range(2)]
    off = _run(dev, model_name, torch.bfloat16, "0", 12, batches, monkeypatch, reassign_at=6)
    on = _run(dev, model_name, torch.bfloat16, "1", 12, batches, monkeypatch, reassign_at=6)
    _assert_equal(off, on)
    states = [v for rep in on[5] for v in rep.values()]
    assert states and all(s.startswith("replaying") for s in states), on[5]


def test_replay_restores_the_stream_when_a_launch_fails(dev, monkeypatch):
    """A launch that fails inside a side-stream section of a replay must leave torch's current stream where the caller had it."""
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import tape, _lib
    from insar_unet_ca_amd.data import make_batch
    monkeypatch.setattr(tape, "MODE", "1")
    x, y = (t.to(dev) for t in make_batch(0, 2, 32))
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.CrossEntropyLoss(ignore_index=255)
    for _ in range(4):
        net.zero_grad(set_to_none=True)
        crit(net(x), y).backward()
    plan = [pl for lst in net._plans.plans.values() for pl in (lst if isinstance(lst, list) else [lst])][0]
    st = plan._tapes[plan._tape_key("b")]
    assert st["state"] == 3
    ops = list(st["tape"])
    first_side = next(i for i, op in enumerate(ops) if op[0] == 1)
    ops[first_side] = (1, lambda *a: -7, ops[first_side][2], "injected failure")
    before = torch.cuda.current_stream()
    with pytest.raises(_lib.InsarError):
        plan._replay(ops, {"dlogits": 0})
    assert torch.cuda.current_stream() == before
    torch.cuda.synchronize()
