"""GPU: the step fed from the host loader (Unet-ChannalAttention.py:339-340, :436-451): pinned batches
copied on a copy stream one batch ahead (data.DevicePrefetcher) must give the training trajectory of device-resident
batches bit for bit, in the loader's order, without ever handing the step a half-copied or recycled buffer."""
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


def _train(dev, feed, steps):
    import insar_unet_ca_amd as iu
    torch.manual_seed(2)
    net = iu.UNet(2, 2, True, compute_dtype=torch.bfloat16).to(dev).train()
    crit = iu.DiceCELoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-3)
    losses = []
    it = iter(feed)
    for _ in range(steps):
        x, y = next(it)
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return [float(l) for l in losses], {k: v.clone() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("pinned", [True, False])
def test_prefetched_batches_give_the_resident_trajectory(dev, pinned):
    import insar_unet_ca_amd as iu
    ds = iu.SyntheticTiles(24, 64)
    loader = iu.make_loader(ds, batch_size=4, shuffle=True, seed=3)          # pin_memory=True, as the reference's loader
    host = [(x.clone(), y.clone()) for x, y in loader]
    assert len(host) == 6
    if pinned:
        host = [(x.pin_memory(), y.pin_memory()) for x, y in host]
    resident = [(x.to(dev), y.to(dev)) for x, y in host]
    steps = 12                                                              # two passes over the six batches

    def cycle(items):
        while True:
            for it in items:
                yield it

    class Twice:                                                            # an iterable the prefetcher can restart
        def __init__(self, items): self.items = items
        def __len__(self): return 2 * len(self.items)
        def __iter__(self): return iter(self.items + self.items)

    l_res, sd_res = _train(dev, cycle(resident), steps)
    l_pre, sd_pre = _train(dev, iu.DevicePrefetcher(Twice(host), dev), steps)
    assert l_res == l_pre
    for k in sd_res:
        assert torch.equal(sd_res[k], sd_pre[k]), k


def test_prefetcher_yields_every_batch_once_in_order(dev):
    """Values, order and exhaustion: 7 distinguishable batches (odd count, last one short) through 2 slots."""
    import insar_unet_ca_amd as iu
    host = [(torch.full((3 if i == 6 else 4, 2, 8, 8), float(i)), torch.full((3 if i == 6 else 4, 8, 8), i, dtype=torch.int64))
            for i in range(7)]
    seen = []
    for x, y in iu.DevicePrefetcher(host, dev):
        # consume on the compute stream with some work in between, as a training step would
        z = (x * 2).sum() / x.numel()
        seen.append((float(z) / 2, int(y.max()), tuple(x.shape)))
    assert seen == [(float(i), i, (3 if i == 6 else 4, 2, 8, 8)) for i in range(7)]


@pytest.mark.parametrize("compact", [True, False])
def test_mask_transport_is_exact(dev, compact):
    """int64 masks in 0..255 cross the link as uint8 (compact_masks) and arrive as the same int64 tensors; a batch with a
    value outside that range (-1, 300) takes the plain path; both interleaved through the same slots."""
    import insar_unet_ca_amd as iu
    g = torch.Generator().manual_seed(7)
    host = []
    for i in range(6):
        y = torch.randint(0, 3, (4, 16, 16), generator=g, dtype=torch.int64)
        y[:, ::5, ::3] = 255                                  # the reference's ignore value
        if i % 3 == 1:
            y[0, 0, 0] = -1                                   # not representable: plain path for this batch
        if i % 3 == 2:
            y[1, 2, 3] = 300
        host.append((torch.randn(4, 2, 16, 16, generator=g), y))
    pf = iu.DevicePrefetcher(host, dev, compact_masks=compact)
    n = 0
    for (xd, yd), (xh, yh) in zip(pf, host):
        assert yd.dtype == torch.int64 and torch.equal(yd.cpu(), yh) and torch.equal(xd.cpu(), xh)
        n += 1
    assert n == 6
