"""Headline benchmark: U-Net-CA training throughput (tiles/s) on synthetic 256x256x2 InSAR tiles.

python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py
A step = one pass of the hot path over one batch: zero_grad -> forward -> Dice+CE loss -> backward
(-> RCCL gradient all-reduce when N > 1) -> Adam step, exactly the loop of
Unet-ChannalAttention.py:342-346. Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# --- algorithmic figures (SURVEY §8d / BASELINE.md) ------------------------------------------------
FLOP_PER_TILE_256 = 288.7e9           # forward + backward, 2x256x256 tile
PEAK_BF16_TFLOPS = 2500.0             # dense MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3               # fp32 MFMA = fp32 vector rate
PEAK_HBM_GBS = 8000.0


def cpu_baseline(size: int = 256):
    """the oracle (CPU restatement of the reference, fp32, torch/oneDNN; pinned to the reference by
    tests/golden) timed on the host cores of this box, on a bounded sample:
      * config 1: eval-mode forward latency of one [1,2,256,256] tile (3 warm-up + 10 timed);
      * the training step of Unet-ChannalAttention.py:342-346 on [4,2,256,256]: CE + Adam(lr=1e-4), 1 warm-up + 3 timed.
    `value` is the training figure (the metric's unit, tiles/s)."""
    from collections import OrderedDict

    from insar_unet_ca_amd.data import make_batch
    from oracle import unet_ca_oracle as orc

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))           # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    import insar_unet_ca_amd as iu

    sd = OrderedDict((k, v.clone()) for k, v in iu.UNet(2, 2, True).state_dict().items())
    x1, _ = make_batch(0, 1, size)
    with torch.no_grad():
        for _ in range(3):
            orc.unet_forward(sd, x1, use_se=True, training=False)
        t0 = time.time()
        for _ in range(10):
            orc.unet_forward(sd, x1, use_se=True, training=False)
        fwd_ms = 1e3 * (time.time() - t0) / 10
    batch = 4
    x, y = make_batch(0, batch, size)
    state = {}
    orc.train_step(sd, state, x, y, use_se=True, lr=1e-4)      # warm-up (primitive creation)
    nsteps = 3
    t0 = time.time()
    for _ in range(nsteps):
        orc.train_step(sd, state, x, y, use_se=True, lr=1e-4)
    dt = time.time() - t0
    return {"value": nsteps * batch / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
            "eval_forward_ms_1x2x256x256": round(fwd_ms, 1), "eval_forward_tiles_per_s": round(1e3 / fwd_ms, 2),
            "sample": f"{nsteps} timed fp32 train steps (fwd + CE + bwd + Adam, Unet-ChannalAttention.py:342-346) of the CPU "
                      f"oracle on {batch} synthetic {size}x{size}x2 tiles each ({dt:.2f} s), and 10 timed eval forwards of one "
                      f"tile (config 1); torch {torch.__version__} CPU/oneDNN, {cores} threads"}


def _traffic_files():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)


def measure(args, world: int, rank: int, dev, dist, iu, engine, make_batch):
    """One configuration: build the model, warm up, time args.steps steps (barrier + synchronize on both sides, max over
    ranks), then the two per-kernel event passes. Returns the result dict on rank 0 (None elsewhere)."""
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.loss is None:
        args.loss = "dice_ce" if args.model == "unet" else "ce"
    channels = 2 if args.model == "unet" else 1
    torch.manual_seed(0)                       # identical random init on every rank
    if args.model == "unet":
        net = iu.UNet(in_channels=2, num_classes=2, use_se=True, compute_dtype=dtype).to(dev).train()
    else:
        net = iu.DeepLabV3_SingleChannel_Attn(num_classes=2, backbone="resnet50", pretrained=False, compute_dtype=dtype).to(dev).train()
    model = net
    if world > 1:
        from insar_unet_ca_amd.parallel import DataParallel
        model = DataParallel(net, shard_optimizer=args.shard_optimizer, bucket_mb=args.bucket_mb)
        model.record_exposed = True            # HIP events around the wait for the gradient exchange (exposed_comm_ms)
    crit = iu.DiceCELoss(ignore_index=255) if args.loss == "dice_ce" else iu.CrossEntropyLoss(ignore_index=255)
    if world > 1 and args.shard_optimizer:
        from insar_unet_ca_amd.parallel import ShardedAdam
        opt = ShardedAdam(model, lr=1e-4, defer_gather=True)
    else:
        opt = iu.Adam(net.parameters(), lr=1e-4)
    adam_fused = world == 1 and args.adam_in_backward == "on" and args.graph != "on"
    if adam_fused:
        opt.fuse_into_backward(net)

    # synthetic tiles, resident in HBM before timing; every rank draws different tiles
    nb = 2
    batches = []
    for b in range(nb):
        x, y = make_batch((rank * nb + b) * args.batch, args.batch, args.size, channels=channels)
        batches.append((x.to(dev), y.to(dev)))

    feed = None
    if args.stream_input:
        # the reference's loop copies every batch from the pinned-memory loader inside the step (:339-340, :436-451); here
        # the host batches (pinned, the same two tiles sets over and over) go through data.DevicePrefetcher: batch i+1 is
        # copied on a copy stream while step i computes. 17 MB of H2D per step at config 2.
        from insar_unet_ca_amd.data import DevicePrefetcher

        class _Cycle:
            def __init__(self, items):
                self.items = items

            def __len__(self):
                return 1 << 30

            def __iter__(self):
                i = 0
                while True:
                    yield self.items[i % len(self.items)]
                    i += 1

        host = [(x.cpu().pin_memory(), y.cpu().pin_memory()) for x, y in batches]
        feed = iter(DevicePrefetcher(_Cycle(host), dev))

    def step(i: int):
        x, y = next(feed) if feed is not None else batches[i % nb]
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x), y)
        loss.backward()
        opt.step()
        return loss

    if os.environ.get("INSAR_MAIN_PRIORITY"):      # diagnostic: the whole step on a stream of this HIP priority
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["INSAR_MAIN_PRIORITY"])))
    use_graph = args.graph == "on" or (args.graph == "auto" and world == 1)
    if use_graph and world > 1:
        print("error: --graph on is not available under data parallelism", file=sys.stderr)
        return 2
    if use_graph and args.stream_input:
        print("error: --stream-input feeds the eager step (the graphed step copies into its own static inputs)", file=sys.stderr)
        return 2
    # Settling: on a box that has just been handed over (the driver's bench is the first long job on its GPU) the first second of
    # steps runs 1-2 % slower than the steady state the metric asks for — plans and launch tapes are being built, clocks and
    # the memory-side cache settle (profiles/r04_flat2.txt, item 7: the first run on a fresh box 6.90 ms, the same command two
    # minutes later 6.79). These untimed steps come BEFORE the W warm-up steps of the contract and are reported as `settle_steps`.
    for i in range(args.settle):
        step(i)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    timed_step = step
    if use_graph:
        gstep = iu.GraphedTrainStep(model, crit, opt, batches[0][0], batches[0][1], warmup=2)

        def timed_step(i: int):
            x, y = batches[i % nb]
            return gstep(x, y)

        for i in range(2):
            timed_step(i)
        torch.cuda.synchronize()

    # the host's own cost of enqueuing a step: three steps into an EMPTY queue (over a whole run the enqueue time also contains
    # the back-pressure of a full command queue, i.e. GPU time: host_enqueue_ms_per_step below saturates at about the GPU time
    # minus ten steps' worth of queue)
    torch.cuda.synchronize()
    th = time.perf_counter()
    for i in range(3):
        timed_step(i)
    host_unblocked = (time.perf_counter() - th) / 3
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        model.exposed_events.clear()
        model.gather_wait_events.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = timed_step(i)
    host_enqueue = time.perf_counter() - t0      # host time to enqueue the K steps (no sync inside a step)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0       # this rank's own K steps, before it meets the others
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dp_info = None
    if world > 1:
        own = own_elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tmin = torch.tensor([own], dtype=torch.float64, device=dev)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        tmax = torch.tensor([own], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ev = model.exposed_events
        exposed = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        t = torch.tensor([exposed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # sharded scheme: time the forward pass's streams sat waiting for parameter buckets (per step: sum over the buckets)
        gw = model.gather_wait_events
        gwait = sum(a.elapsed_time(b) for a, b in gw) / max(args.steps, 1)
        tg = torch.tensor([gwait], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        dp_info = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                   "exchange": "reduce_scatter+sharded_adam+all_gather" if args.shard_optimizer else "allreduce",
                   "bucket_mb": args.bucket_mb, "buckets": len(model.sharded.bounds) if model.sharded is not None else None,
                   # max over ranks of the mean time the main stream sat waiting for the gradient exchange after the last
                   # backward kernel (HIP events around DataParallel's wait; 0 = fully overlapped with backward)
                   "exposed_comm_ms": round(float(t.item()) + float(tg.item()), 3),
                   "exposed_gradient_exchange_ms": round(float(t.item()), 3),
                   # sharded scheme only: the side stream's waits for parameter buckets at the start of forward (they gate a
                   # stage's first launch through an event: an upper bound of what the main stream actually waited)
                   "exposed_param_allgather_ms": round(float(tg.item()), 3) if args.shard_optimizer else None,
                   # a rank's own K steps up to its device synchronise (before the closing barrier): fastest and slowest rank
                   "rank_ms_per_step_min": round(1e3 * float(tmin.item()) / args.steps, 3),
                   "rank_ms_per_step_max": round(1e3 * float(tmax.item()) / args.steps, 3)}
        model.record_exposed = False

    # Roofline leg: the same K steps twice more, in the same process, with a HIP-event pair around every GEMM-class
    # launch, recorded on the stream the kernel is launched on (main or side). Pass 1 keeps the launch configuration
    # of the timed region (weight gradients on the side stream beside the dgrad chain, split-K for engine.WGRAD_FILL of the work-group slots);
    # pass 2 runs everything on ONE stream (each kernel alone on the chip, split-K that fills it). Both are kept out
    # of the timed region above because ~130 event records per step cost host time and perturb the overlap.
    timers = {}
    if not args.no_kernel_timing:
        for mode in ("timed_config", "alone"):
            timer = engine.KernelTimer(alone=(mode == "alone"))
            engine.PROFILER = timer
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            torch.cuda.synchronize()
            timer.elapsed = time.perf_counter() - t1
            engine.PROFILER = None
            timers[mode] = timer
    final_loss = float(loss.detach())

    if rank == 0:
        tiles = args.batch * world * args.steps
        value = tiles / elapsed
        ms = 1e3 * elapsed / args.steps
        per_gpu = value / world
        if args.model == "unet":
            name, flop_tile = "U-Net-CA (in=2, classes=2, use_se)", FLOP_PER_TILE_256 * (args.size / 256) ** 2
            act_bytes = (754e6 if args.dtype == "bf16" else 1508e6) * (args.size / 256) ** 2
            nparams = 31261122
        else:
            # DeepLabV3-CA: algorithmic FLOPs from the layer shapes of the plan (every tap of every convolution, forward +
            # input gradient + weight gradient; the 1-channel stem has no input gradient) and algorithmic BYTES by the rules
            # SURVEY 8d states for the U-Net (ideal fusion): a convolution reads its input and writes its output once,
            # BatchNorm / ReLU / the residual add / the attention scale cost nothing extra (applied on load), backward per
            # conv + BN unit reads the saved input, writes the input gradient and reads (grad-out, raw-out) once per inherent
            # reduction pass (two: BatchNorm-backward sums; dgrad / wgrad) => 3 in + 5 out elements per unit; MaxPool(3,2,1)
            # forward in + out, backward in + g_out + g_in; the channel-attention module one extra read of its input forward
            # and backward; plus per step 36 bytes per parameter (bf16 shadow + fp32 master + gradient + Adam).
            plan = next(iter(net._plans.plans.values()))[0]
            macs = sum(u.M * u.cin * u.cout * u.k * u.k for u in plan.units) / args.batch
            flop_tile = 3 * 2.0 * macs + 2 * 2.0 * (args.size // 2) ** 2 * 64 * 49
            elems = 0.0
            for u in plan.units:
                m_in = u.x.B * u.x.H * u.x.W
                elems += 3.0 * m_in * u.cin + 5.0 * u.M * u.cout
            q = (args.size // 4) ** 2 * 64 * args.batch          # MaxPool(3,2,1) output elements (stem: size/2, 64 channels in)
            elems += (4 * q + q) + (3 * 4 * q)                    # pool forward in + out, backward in + g_out + g_in (in = 4 q)
            elems += 2.0 * args.batch * (args.size // 8) ** 2 * 256   # ChannelAttentionModule: one extra read each way
            name, nparams = "DeepLabV3-CA (ResNet-50 os8 + ASPP + CAM, in=1, classes=2)", 39635906
            act_bytes = elems * 2 / args.batch
        out = {
            "metric": "InSAR tiles/sec (fwd+bwd) U-Net-CA 256x256" if args.model == "unet" else "InSAR tiles/sec (fwd+bwd) DeepLabV3-CA 256x256 (config 5)",
            "value": round(value, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": args.settle, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{name} {args.dtype}, batch {args.batch}x{channels}x{args.size}x{args.size} "
                                   f"per GPU, {'Dice+CE' if args.loss == 'dice_ce' else 'CE'} + Adam(lr=1e-4) training on synthetic InSAR tiles",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "final_loss": round(final_loss, 5), "hipgraph": bool(use_graph),
            "optimizer": ("Adam(lr=1e-4), update + weight re-layout issued stage by stage inside backward on the side stream "
                          "(Adam.fuse_into_backward; bitwise the plain step)" if adam_fused else "Adam(lr=1e-4), optimizer.step() after backward"), "input": "streamed from pinned host memory (copy stream, one batch ahead)" if args.stream_input else "resident in HBM",
            "host_enqueue_ms_per_step": round(1e3 * host_enqueue / args.steps, 3),
            "host_ms_per_step_empty_queue": round(1e3 * host_unblocked, 3),
            "hbm_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
        }
        try:        # launch tapes (insar_unet_ca_amd/tape.py): which of the plan's passes were replayed from a recorded launch list
            plans = [pl for lst in net._plans.plans.values() for pl in lst]
            out["launch_tape"] = {str(k): v for pl in plans for k, v in pl.tape_report().items()}
        except Exception as e:      # noqa: BLE001 - reporting only
            out["launch_tape"] = f"unavailable: {e}"
        if dp_info is not None:
            out.update(dp_info)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        per_param = 36 if args.dtype == "bf16" else 40      # weights fwd + bwd, gradient write, Adam
        bytes_tile = (act_bytes + per_param * nparams / args.batch) if act_bytes is not None else 0.0
        fl = per_gpu * flop_tile / 1e12
        bw = per_gpu * bytes_tile / 1e9
        out["frac_of_mfma_roofline"] = round(fl / peak, 4)
        out["frac_of_hbm_roofline"] = round(bw / PEAK_HBM_GBS, 4)
        bound = "mfma" if fl / peak >= bw / PEAK_HBM_GBS else "hbm"
        out["roofline_step"] = {"bound": bound, "achieved": round(fl if bound == "mfma" else bw, 2),
                                "peak": peak if bound == "mfma" else PEAK_HBM_GBS, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                                "frac": round(max(fl / peak, bw / PEAK_HBM_GBS), 4),
                                "algorithmic": (f"{flop_tile / 1e9:.1f} GFLOP and {bytes_tile / 1e6:.0f} MB per tile "
                                                + ("(SURVEY 8d)" if args.model == "unet" else "(layer shapes of the plan, SURVEY 8d's ideal-fusion rules)")
                                                + ", whole step incl. loss and Adam, timed region"),
                                "frac_mfma": round(fl / peak, 4), "frac_hbm": round(bw / PEAK_HBM_GBS, 4)}
        if args.model == "unet" and args.dtype == "bf16" and args.size == 256 and args.batch == 16:
            # whole-step HBM traffic from the PMC passes of this configuration (tools/pmc_traffic.py: every dispatch between
            # two Adam launches), next to the algorithmic bytes of the byte model
            for tpath in _traffic_files():
                try:
                    st = json.load(open(tpath)).get("__step__")
                except Exception:
                    st = None
                if st:
                    alg = bytes_tile * args.batch
                    out["roofline_step"]["traffic"] = st["hbm_bytes_per_step"]
                    out["roofline_step"]["algorithmic_bytes_per_step"] = round(alg)
                    out["roofline_step"]["traffic_over_algorithmic"] = round(st["hbm_bytes_per_step"] / alg, 3)
                    out["roofline_step"]["traffic_source"] = os.path.relpath(tpath, ROOT)
                    break
        if timers:
            summ = timers["timed_config"].summary()
            alone = timers["alone"].summary()
            t_el = timers["timed_config"].elapsed
            dom_tag = max(summ, key=lambda k: summ[k]["ms"]) if summ else None
            if dom_tag:
                dom = summ[dom_tag]
                traffic, tfile, tconf = None, None, None
                for tpath in _traffic_files():
                    try:
                        tj = json.load(open(tpath))
                    except Exception:
                        continue
                    stem = dom_tag.split(" +")[0].rstrip(">")          # the profiler's name carries further template arguments
                    hit = [v for k, v in tj.items() if k.startswith(stem) and isinstance(v, dict) and "hbm_bytes_per_launch" in v]
                    if hit:
                        traffic = hit[0]["hbm_bytes_per_launch"]
                        tfile = os.path.relpath(tpath, ROOT)
                        tconf = tj.get("__config__", {}).get("launch_configuration")
                        break
                al = alone.get(dom_tag)
                out["roofline"] = {
                    "kernel": dom_tag, "bound": "mfma", "achieved": round(dom["tflops"], 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4), "traffic": traffic,
                    "traffic_launch_configuration": tconf,
                    "flop_per_launch": round(dom["flops"] / dom["launches"], 1),
                    "algorithmic_bytes_per_launch": round(dom.get("bytes", 0.0) / dom["launches"], 1) or None,
                    "avg_launch_us": round(dom["avg_us"], 2), "launches": dom["launches"],
                    "share_of_step": round(dom["ms"] / (1e3 * t_el), 4),
                    "alone": ({"achieved": round(al["tflops"], 2), "frac": round(al["tflops"] / peak, 4),
                               "avg_launch_us": round(al["avg_us"], 2)} if al else None),
                    "measured": "HIP events around each launch on the stream it is launched on, in a second pass of the "
                                "same %d steps in the launch configuration of the timed region (weight gradients on the "
                                f"side stream beside the dgrad chain, split-K for {engine.WGRAD_FILL:g} of the work-group slots: kernels that "
                                "overlap share the chip, so their durations are longer than alone; %.2f ms/step with "
                                "events vs %.2f ms/step timed); `alone` = a third pass on ONE stream, every kernel by "
                                "itself on the chip (%.2f ms/step); compare profiles/ kernel stats; traffic = rocprofv3 "
                                "PMC (2*FETCH_SIZE + WRITE_SIZE) per launch from %s, taken in the launch configuration named "
                                "by traffic_launch_configuration (algorithmic_bytes_per_launch is the timed configuration's)" %
                                (args.steps, 1e3 * t_el / args.steps, ms,
                                 1e3 * timers["alone"].elapsed / args.steps, tfile or "profiles/ (none found)"),
                }

            def table(sm, el):
                return {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 2),
                            "share_of_step": round(v["ms"] / (1e3 * el), 4)}
                        for k, v in sorted(sm.items(), key=lambda kv: -kv[1]["ms"])}

            def total(sm, el):
                gm, gf = sum(v["ms"] for v in sm.values()), sum(v["flops"] for v in sm.values())
                return {"tflops": round(gf / (gm * 1e-3) / 1e12, 2) if gm > 0 else 0.0, "ms_per_step": round(gm / args.steps, 3),
                        "share_of_step": round(gm / (1e3 * el), 4)}

            out["gemm_kernels"] = table(summ, t_el)
            out["gemm_total"] = total(summ, t_el)
            out["gemm_kernels_alone"] = table(alone, timers["alone"].elapsed)
            out["gemm_total_alone"] = total(alone, timers["alone"].elapsed)
            plain = {k: v for k, v in alone.items() if "+bstat" not in k}
            if len(plain) != len(alone):
                # launches tagged "+bstat" also do a BatchNorm-backward reduce pass's work in their epilogue (they read the
                # consumer unit's y and fold two sums per channel: InsarBstat); the same kernels without it:
                out["gemm_total_alone_plain_launches"] = total(plain, timers["alone"].elapsed)
        return out
    return None


def other_configs(args, dev, dist, iu, engine, make_batch) -> dict:
    """Configs 4 and 5 of BASELINE.json for the same number of steps, in the same process, so that the driver's one bench
    line records them beside the headline (config 2). Same timed-region rules (measure()); no per-kernel event passes."""
    import copy
    import gc
    res = {}
    for key, over in (("config4_unet_fp32_512_b8_ce", dict(dtype="f32", size=512, batch=8, loss="ce", model="unet")),
                      ("config5_deeplabv3_ca_bf16_256_b16_ce", dict(dtype="bf16", size=256, batch=16, loss="ce", model="deeplab"))):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        a.no_kernel_timing, a.graph, a.stream_input = True, "off", False
        a.settle = min(args.settle, 20)
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        r = measure(a, 1, 0, dev, dist, iu, engine, make_batch)
        res[key] = ({k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "settle_steps", "dtype", "config", "roofline_step",
                                       "host_enqueue_ms_per_step", "hbm_allocated_gb", "final_loss") if k in r}
                    if isinstance(r, dict) else {"error": f"measure() returned {r!r}"})
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 timed steps after 20 (0.8 s of GPU time). 20-step runs scatter by +-0.04 ms/step on one box and read
    # 0.05-0.07 ms higher than 150- / 400-step runs (profiles/r04_flat2.txt, item 7): clocks and caches are still settling
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed steps before the --warmup steps (plan / tape construction, clocks; reported as settle_steps; 0 = none)")
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU (config 2: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--loss", default=None, choices=["dice_ce", "ce"], help="default: dice_ce for unet (config 2), ce for deeplab")
    ap.add_argument("--model", default="unet", choices=["unet", "deeplab"],
                    help="unet = U-Net-CA (configs 2-4, the headline metric); deeplab = DeepLabV3-CA (config 5, 1-channel tiles)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: reduce-scatter + Adam on the 1/N shard + all-gather instead of all-reduce + full Adam")
    ap.add_argument("--graph", default="off", choices=["auto", "on", "off"],
                    help="replay the step from a captured hipGraph (insar_unet_ca_amd.GraphedTrainStep): off by default (measured slower than eager launches on ROCm 7.2: 8.6 vs 7.9 ms/step); auto = on for 1 GPU, off "
                         "under data parallelism (the RCCL collectives inside backward are issued eagerly)")
    ap.add_argument("--adam-in-backward", default="off", choices=["on", "off"],
                    help="1 GPU: Adam.fuse_into_backward — the optimizer update and the weight re-layout of a backward stage run on the "
                         "side stream as soon as the stage's gradients are enqueued (same arithmetic, bitwise the plain step). Measured "
                         "SLOWER (7.77 -> 8.29 ms/step: 875 MB of optimizer traffic beside the dgrad chain costs more than the 0.17 ms "
                         "it hides), so off by default: optimizer.step() after backward does all of it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip configs 4 and 5 (the default 1-GPU run of config 2 reports them under `other_configs`)")
    ap.add_argument("--bucket-mb", type=float, default=16.0, help="N > 1: minimum size of a gradient-exchange bucket (MiB)")
    ap.add_argument("--allow-switches", action="store_true",
                    help="measure although INSAR_* environment switches that select kernels / launch paths are set to non-default "
                         "values (A/B runs); without it such a run exits with code 3. The line reports them under `switches` either way")
    ap.add_argument("--wgrad-fill", type=float, default=None,
                    help="share of the work-group slots a side-stream weight gradient aims at (engine.WGRAD_FILL, default 0.5; the "
                         "transposed convs' 0.7 scales with it): the 8-GPU operator's knob for trading side-queue CUs against RCCL's")
    ap.add_argument("--stream-input", action="store_true",
                    help="feed every step from the host: batches in pinned memory, copied on a copy stream while the previous "
                         "step computes (the reference copies per step, Unet-ChannalAttention.py:339-340)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
    ndev = torch.cuda.device_count()
    if world > ndev and args.backend == "nccl":
        # one process per GPU over RCCL: folding ranks onto one device would silently measure something else
        print(f"error: WORLD_SIZE={world} but only {ndev} GPU(s) visible; RCCL needs one device per rank "
              "(use --backend gloo for a functional rehearsal on fewer GPUs)", file=sys.stderr)
        return 2
    local_rank = local_rank % max(ndev, 1)          # gloo rehearsal only: several ranks may share the one visible GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist

    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine, switches
    from insar_unet_ca_amd.data import make_batch

    # every INSAR_* variable that is set to something other than its default goes into the line; a default run refuses to
    # measure with a kernel-selecting one set (a stray variable on a box would otherwise change the measurement without trace)
    sw = switches.non_default([os.path.abspath(__file__)])
    selecting = [k for k, v in sw.items() if v["kernel_selecting"]]
    if selecting and not args.allow_switches:
        print(f"error: non-default kernel-selecting switches in the environment: {', '.join(selecting)} "
              "(unset them, or pass --allow-switches for an A/B run)", file=sys.stderr)
        return 3
    if args.wgrad_fill is not None:
        if not 0.05 <= args.wgrad_fill <= 1.0:
            print("error: --wgrad-fill must lie in [0.05, 1]", file=sys.stderr)
            return 2
        engine.WGRAD_FILL_T = min(1.0, engine.WGRAD_FILL_T * args.wgrad_fill / engine.WGRAD_FILL)
        engine.WGRAD_FILL_SMALL = min(1.0, engine.WGRAD_FILL_SMALL * args.wgrad_fill / engine.WGRAD_FILL)
        engine.WGRAD_FILL = args.wgrad_fill

    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    out = measure(args, world, rank, dev, dist, iu, engine, make_batch)
    if isinstance(out, int):
        return out
    if rank == 0 and isinstance(out, dict):
        out["switches"] = {k: v["value"] for k, v in sw.items()}
        out["wgrad_fill"] = {"conv3x3": engine.WGRAD_FILL, "conv3x3_small_tiles": engine.WGRAD_FILL_SMALL, "conv_transpose": engine.WGRAD_FILL_T, "deeplab": engine.WGRAD_FILL_DL}
        if world == 1 and not args.no_other_configs and args.model == "unet" and args.dtype == "bf16" and args.size == 256:
            out["other_configs"] = other_configs(args, dev, dist, iu, engine, make_batch)
        if world == 1 and not args.no_cpu_baseline and args.model == "unet":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
