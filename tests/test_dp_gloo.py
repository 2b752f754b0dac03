"""CPU-only, world_size 2 over gloo: the data-parallel gradient exchange. Each rank fills the flat
gradient buffer with the ORACLE gradients of its own shard (local-BN semantics) and runs the same
BucketReducer bench.py / DataParallel use under RCCL; the result must be the mean of the per-shard
gradients recorded in the golden fixture g9 (generated from the reference itself)."""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank: int, world: int, port: int, out_path: str):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from insar_unet_ca_amd.parallel import BucketReducer, plan_buckets
    from oracle import closed_form as cf
    from oracle import unet_ca_oracle as orc

    sd = cf.fill_state_dict(orc.state_dict_template(2, 2, True))
    names = [k for k in sd if orc.is_param(k)]
    work = OrderedDict(sd)
    leaves = []
    for k in names:
        work[k] = sd[k].clone().requires_grad_(True)
        leaves.append(work[k])
    x = cf.make_input((2, 2, 32, 32), salt=1.1 * rank)
    loss = orc.cross_entropy(orc.unet_forward(work, x, True, True), cf.make_target((2, 32, 32)))
    grads = torch.autograd.grad(loss, leaves)
    # flat buffer in an arbitrary "completion order" (reverse registration), bucketed like the engine does
    order = list(reversed(range(len(names))))
    sizes = [grads[i].numel() for i in order]
    flat = torch.cat([grads[i].reshape(-1) for i in order])
    ends = np.cumsum(sizes)
    closes = plan_buckets(sizes, 1 << 20)
    red = BucketReducer()
    begin = 0
    for stage in range(len(sizes)):
        if stage in closes:
            red.reduce_slice(flat, begin, int(ends[stage]))
            begin = int(ends[stage])
    red.finish()
    if rank == 0:
        out, off = {}, 0
        for i, n in zip(order, sizes):
            out[names[i]] = flat[off:off + n].view(grads[i].shape).clone()
            off += n
        torch.save(out, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_mean_of_shard_oracle_grads(golden, tmp_path):
    from tests.helpers import check_summary
    port = 29500 + (os.getpid() % 2000)
    out_path = str(tmp_path / "reduced.pt")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    reduced = torch.load(out_path)
    g9 = golden("g9_dp")
    assert len(reduced) == 100
    for k, v in reduced.items():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            # a conv bias in front of a training-mode BatchNorm has an exactly-zero gradient; what torch
            # reports is summation noise (~1e-9) that changes with the thread count, not a value to match
            assert float(v.abs().max()) < 1e-6
            continue
        check_summary(g9, f"mean_grad/{k}", v, 2e-4)


# ------------------------------------------------------------------------------------------------------------
# world 8: sampler + bucket plan + reducer on the engine's real flat-buffer layout
# ------------------------------------------------------------------------------------------------------------
class _FakeCtx:
    """engine.Ctx's stream interface on a host without a GPU (everything runs in order on the CPU)."""

    def side_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def join_side(self):
        pass


class _FakeSink:
    def __init__(self, flat):
        self._flat = flat

    def flat(self):
        return self._flat


class _FakePlan:
    """What DataParallel's hooks read from an engine.UNetPlan: the stage layout of the flat gradient buffer."""

    def __init__(self, net, flat):
        from insar_unet_ca_amd import engine
        from insar_unet_ca_amd.parallel import plan_buckets
        self.groups = engine.grad_groups(net)
        self.offs, self.stage_sizes, self.total = engine.flat_layout(self.groups)
        self.stage_ends = list(np.cumsum(self.stage_sizes))
        self.ctx, self.sink = _FakeCtx(), _FakeSink(flat)
        self._pb = plan_buckets

    def bucket_closes(self, min_elems):
        return set(self._pb(self.stage_sizes, min_elems))


def _run_backward_hooks(net, plan):
    """The call sequence of modules._UNetFn.backward around engine.UNetPlan.backward (9 stages)."""
    h = net._hooks
    h["on_begin"](plan)
    for stage in range(len(plan.stage_sizes)):
        h["on_bucket"](plan, ("stage", stage))
    h["on_done"](plan)


def _cpu_adam_rows(rows, lr, b1, b2, eps, bc1, bc2_sqrt):
    """Element-wise Adam with the kernel's formula (csrc/loss_optim.hip adam_kernel = torch.optim.Adam, :466): test
    stand-in for the HIP launch so the EXCHANGE logic of both schemes can run under gloo on a host without a GPU."""
    for p, g, m, v in rows:
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(m, v.sqrt().div_(bc2_sqrt).add_(eps), value=-lr / bc1)


def _worker8(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import ShardedSampler, SyntheticTiles
    from insar_unet_ca_amd.parallel import DataParallel
    from oracle import closed_form as cf
    from oracle import unet_ca_oracle as orc

    torch.manual_seed(1000 + rank)                         # different init per rank: the wrapper's broadcast fixes it
    net = iu.UNet(2, 2, True)
    dp = DataParallel(net, bucket_mb=8.0)
    sd = OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items())
    # this rank's tiles: ShardedSampler deals the 16-tile global batch round-robin (rank r takes r::8)
    ds = SyntheticTiles(16, 16, offset=4000)
    idx = list(ShardedSampler(len(ds), rank, world, shuffle=False))
    assert idx == list(range(rank, 16, world))
    x = torch.stack([ds[i][0] for i in idx])
    y = torch.stack([ds[i][1] for i in idx])
    names = [k for k in sd if orc.is_param(k)]
    work = OrderedDict(sd)
    for k in names:
        work[k] = sd[k].clone().requires_grad_(True)
    loss = orc.cross_entropy(orc.unet_forward(work, x, True, True), y)
    grads = dict(zip(names, torch.autograd.grad(loss, [work[k] for k in names])))
    # lay the local gradients out exactly as engine.GradSink does, then run the wrapper's backward hooks
    by_param = {id(p): k for k, p in net.named_parameters()}
    plan = _FakePlan(net, None)
    flat = torch.zeros(plan.total)
    for p, o in zip([q for g in plan.groups for q in g], plan.offs):
        flat[o:o + p.numel()] = grads[by_param[id(p)]].reshape(-1)
    local = flat.clone()
    plan.sink = _FakeSink(flat)
    _run_backward_hooks(net, plan)
    # expected mean from a strided sample of every rank's LOCAL buffer (1/97 of 31 M elements: the gather stays small);
    # it walks every parameter tensor and the padding between stages
    sample = local[::97].contiguous()
    gathered = [torch.zeros_like(sample) for _ in range(world)] if rank == 0 else None
    dist.gather(sample, gathered, dst=0)
    if rank == 0:
        mean = torch.stack(gathered).double().mean(0)
        torch.save({"reduced": flat[::97].clone(), "mean": mean, "buckets": len(plan.bucket_closes(dp.min_elems))},
                   os.path.join(out_dir, "w8.pt"))
    # every rank ends with the same buffer: compare checksums
    chk = torch.tensor([float(flat.double().sum()), float(flat[::1013].double().abs().sum())], dtype=torch.float64)
    allchk = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allchk, chk)
    assert all(torch.equal(c, allchk[0]) for c in allchk)
    dist.barrier()
    dist.destroy_process_group()


def test_world8_bucketed_allreduce_on_the_engine_layout(tmp_path):
    """8 ranks (gloo): parameters broadcast from rank 0, tiles dealt by ShardedSampler, each rank's ORACLE gradients
    laid out as engine.GradSink does, exchanged by DataParallel's own hooks in >= 8 MiB buckets: every rank must end
    with the mean of the 8 shard gradients (SURVEY 8e's DP parity oracle), also in the padding between stages."""
    port = 31000 + (os.getpid() % 2000)
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    r = torch.load(tmp_path / "w8.pt")
    assert r["buckets"] >= 3
    scale = float(r["mean"].abs().max())
    assert scale > 0
    assert float((r["reduced"].double() - r["mean"]).abs().max()) <= 2e-6 * scale


# ------------------------------------------------------------------------------------------------------------
# world 2: reduce-scatter + sharded Adam + all-gather == all-reduce + Adam, bit for bit
# ------------------------------------------------------------------------------------------------------------
def _worker_sharded(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd.parallel import DataParallel, ShardedAdam

    def run(shard: bool):
        torch.manual_seed(5)
        net = iu.UNet(2, 2, True)
        dp = DataParallel(net, bucket_mb=8.0, shard_optimizer=shard)
        plan = _FakePlan(net, None)
        params = [q for g in plan.groups for q in g]
        m = torch.zeros(plan.total)
        v = torch.zeros(plan.total)
        opt = ShardedAdam(dp, lr=1e-3, kernel=_cpu_adam_rows, defer_gather=(world == 2)) if shard else None
        for step in range(3):
            if step == 2:
                # resume / re-initialisation AFTER wrapping: an in-place load must be what the next step starts
                # from. A rank shard that was a stale copy of the flat buffer would silently revert it.
                if shard:
                    dp.params_ready()
                sd_new = OrderedDict((k, (v * 0.5 + 0.01) if v.is_floating_point() else v) for k, v in net.state_dict().items())
                net.load_state_dict(sd_new)
            gen = torch.Generator().manual_seed(77 * step + rank)
            flat = torch.randn(plan.total, generator=gen) * 1e-2
            plan.sink = _FakeSink(flat)
            if shard and world == 2 and dp._gathers:
                # what engine.UNetPlan._stage_gates does at the start of the next forward: wait bucket by bucket, in forward
                # order of the backward stages; a bucket is only waited for when the first stage that needs it comes up
                nst = len(plan.stage_sizes)
                inflight = set(dp._gathers)
                assert inflight == set(range(len(dp.sharded.bounds)))          # deferred: every bucket still in flight
                seen = []
                for st in reversed(range(nst)):
                    net._hooks["param_waits"](st)
                    seen.append(set(dp._gathers))
                    assert dp._stage_bucket[st] not in dp._gathers
                    assert all(b in dp._gathers for b in inflight if b < dp._stage_bucket[st])   # later-needed buckets keep flying
                assert not dp._gathers
            _run_backward_hooks(net, plan)
            if shard:
                opt.step()
            else:           # the same arithmetic on every element of every parameter, as optim.Adam does
                b1, b2, t = 0.9, 0.999, step + 1
                rows = [(p.data.view(-1), flat[o:o + p.numel()], m[o:o + p.numel()], v[o:o + p.numel()])
                        for p, o in zip(params, plan.offs)]
                _cpu_adam_rows(rows, 1e-3, b1, b2, 1e-8, 1 - b1 ** t, (1 - b2 ** t) ** 0.5)
        return OrderedDict((k, p.detach().clone()) for k, p in net.named_parameters()), dp

    plain, _ = run(False)
    sharded_run = run(True)
    sharded_run[1].params_ready()
    sharded, dp = OrderedDict((k, p.detach().clone()) for k, p in sharded_run[1].module.named_parameters()), sharded_run[1]
    assert all(p.grad is None for p in dp.module.parameters())
    assert all(sh.data_ptr() == dp.flat_p.data_ptr() + 4 * (b + rank * sh.numel())
               for (b, e), sh in zip(dp.sharded.bounds, dp.sharded.p))          # the shard IS the flat buffer's slice
    # state_dict still has the reference's 154 keys and the parameters are views of the flat buffer
    assert len(dp.module.state_dict()) == 154
    base, n = dp.flat_p.data_ptr(), dp.flat_p.numel() * 4
    assert all(base <= p.data_ptr() < base + n for p in dp.module.parameters())
    same = all(torch.equal(plain[k], sharded[k]) for k in plain)
    moved = any(not torch.equal(plain[k], torch.zeros_like(plain[k])) for k in plain)
    torch.save({"same": same, "moved": moved, "w": sharded["outc.weight"], "shard_elems": sum(t.numel() for t in dp.sharded.p),
                "total": dp.flat_p.numel()}, os.path.join(out_dir, f"sh{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_adam_equals_allreduce_adam_bitwise(tmp_path, world):
    """reduce-scatter -> Adam on the 1/world shard -> in-place all-gather gives the parameters of all-reduce + Adam bit
    for bit, at world 2 (deferred all-gather waits) and at world 8 (the 8-GPU node's rank count), including a
    load_state_dict between steps."""
    port = 33000 + (os.getpid() % 2000) + world
    mp.spawn(_worker_sharded, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"sh{r}.pt") for r in range(world)]
    assert all(r["same"] for r in rs) and rs[0]["moved"]
    assert all(torch.equal(rs[0]["w"], r["w"]) for r in rs)
    assert rs[0]["shard_elems"] * world == rs[0]["total"]           # each rank holds (and updates) 1/world of the optimizer state
