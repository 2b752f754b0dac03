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
BYTES_PER_TILE_BF16_B16 = 825e6       # ideal-fusion HBM traffic per tile at B=16 (bf16 activations)
PEAK_BF16_TFLOPS = 2500.0             # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def cpu_baseline(batch: int = 8, size: int = 256):
    """The oracle (CPU restatement of the reference's training step, fp32, torch/oneDNN) timed on the
    host cores of this box: one untimed + one timed step on a bounded sample."""
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
    x, y = make_batch(0, batch, size)
    state = {}
    orc.train_step(sd, state, x, y, use_se=True, lr=1e-4, dice_weight=1.0)      # warm-up (primitive creation)
    nsteps = 3
    t0 = time.time()
    for _ in range(nsteps):
        orc.train_step(sd, state, x, y, use_se=True, lr=1e-4, dice_weight=1.0)
    dt = time.time() - t0
    return {"value": nsteps * batch / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} timed fp32 train steps (fwd+Dice/CE+bwd+Adam) of the CPU oracle on {batch} synthetic "
                      f"{size}x{size}x2 tiles each, torch {torch.__version__} CPU, {cores} threads; {dt:.2f} s"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU (config 2: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--loss", default="dice_ce", choices=["dice_ce", "ce"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)          # rehearsal: several ranks may share the one visible GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist

    import insar_unet_ca_amd as iu
    from insar_unet_ca_amd import engine
    from insar_unet_ca_amd.data import make_batch

    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)                       # identical random init on every rank
    net = iu.UNet(in_channels=2, num_classes=2, use_se=True, compute_dtype=dtype).to(dev).train()
    model = net
    if world > 1:
        from insar_unet_ca_amd.parallel import DataParallel
        model = DataParallel(net)
    crit = iu.DiceCELoss(ignore_index=255) if args.loss == "dice_ce" else iu.CrossEntropyLoss(ignore_index=255)
    opt = iu.Adam(net.parameters(), lr=1e-4)

    # synthetic tiles, resident in HBM before timing; every rank draws different tiles
    nb = 2
    batches = []
    for b in range(nb):
        x, y = make_batch((rank * nb + b) * args.batch, args.batch, args.size)
        batches.append((x.to(dev), y.to(dev)))

    def step(i: int):
        x, y = batches[i % nb]
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x), y)
        loss.backward()
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    host_enqueue = time.perf_counter() - t0      # host time to enqueue the K steps (no sync inside a step)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Roofline leg: the same K steps again, in the same process, with a HIP-event pair around every
    # GEMM-class launch (recorded on the stream the kernels run on). Kept out of the timed region above
    # because ~130 event records per step cost ~2.5 ms/step (they serialise the queue).
    timer = None
    timed_elapsed = None
    if not args.no_kernel_timing and rank == 0:
        timer = engine.KernelTimer()
        engine.PROFILER = timer
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        timed_elapsed = time.perf_counter() - t1
        engine.PROFILER = None
    elif not args.no_kernel_timing:
        for i in range(args.steps):         # keep the other ranks in lock-step with rank 0's extra pass
            step(i)
        torch.cuda.synchronize()
    final_loss = float(loss.detach())

    if rank == 0:
        tiles = args.batch * world * args.steps
        value = tiles / elapsed
        ms = 1e3 * elapsed / args.steps
        per_gpu = value / world
        out = {
            "metric": "InSAR tiles/sec (fwd+bwd) U-Net-CA 256x256", "value": round(value, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"U-Net-CA (in=2, classes=2, use_se) {args.dtype}, batch {args.batch}x2x{args.size}x{args.size} "
                                   f"per GPU, {'Dice+CE' if args.loss == 'dice_ce' else 'CE'} + Adam(lr=1e-4) training on synthetic InSAR tiles",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "final_loss": round(final_loss, 5),
            "host_enqueue_ms_per_step": round(1e3 * host_enqueue / args.steps, 3),
            "hbm_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
            "frac_of_mfma_roofline": round(per_gpu * FLOP_PER_TILE_256 * (args.size / 256) ** 2 / (PEAK_BF16_TFLOPS * 1e12), 4),
            "frac_of_hbm_roofline": round(per_gpu * BYTES_PER_TILE_BF16_B16 * (args.size / 256) ** 2 / (PEAK_HBM_GBS * 1e9), 4),
        }
        if timer is not None:
            summ = timer.summary()
            dom_tag = max(summ, key=lambda k: summ[k]["ms"]) if summ else None
            if dom_tag:
                dom = summ[dom_tag]
                peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
                traffic = None
                tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
                if os.path.isfile(tpath):
                    try:
                        traffic = json.load(open(tpath)).get(dom_tag, {}).get("hbm_bytes_per_launch")
                    except Exception:
                        traffic = None
                out["roofline"] = {
                    "kernel": dom_tag, "bound": "mfma", "achieved": round(dom["tflops"], 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4), "traffic": traffic,
                    "flop_per_launch": round(dom["flops"] / dom["launches"], 1),
                    "algorithmic_bytes_per_launch": round(dom.get("bytes", 0.0) / dom["launches"], 1) or None,
                    "avg_launch_us": round(dom["avg_us"], 2), "launches": dom["launches"],
                    "share_of_step": round(dom["ms"] / (1e3 * timed_elapsed), 4),
                    "measured": "HIP events around each launch on the launch stream, in a second pass of the same %d "
                                "steps run on ONE stream (as with INSAR_SIDE_STREAM=0: per-kernel durations without "
                                "the weight-gradient overlap, weight gradients with the split-K factor that fills "
                                "the chip; in the timed region they run beside the dgrad chain on half the slots; "
                                "%.2f ms/step with events vs %.2f ms/step in the timed "
                                "region); compare profiles/r01_bench_kernel_stats_single_stream.csv; traffic = "
                                "rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) per launch from profiles/r01_pmc_traffic.json" %
                                (args.steps, 1e3 * timed_elapsed / args.steps, ms),
                }
            gemm_ms = sum(v["ms"] for v in summ.values())
            gemm_fl = sum(v["flops"] for v in summ.values())
            out["gemm_kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                                       "tflops": round(v["tflops"], 2), "share_of_step": round(v["ms"] / (1e3 * timed_elapsed), 4)}
                                   for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
            out["gemm_total"] = {"tflops": round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 2),
                                 "share_of_step": round(gemm_ms / (1e3 * timed_elapsed), 4)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
