"""Synthetic InSAR tiles: wrapped interferometric phase with linear deformation
features, delivered as (cos phi, sin phi) channels in [-1, 1] — the range the reference's
Normalize(0.5, 0.5) produces (Unet-ChannalAttention.py:431) — plus a {0,1} int64 label that marks
pixels within 2 px of a feature line. Seeded with numpy PCG64 only (no torch RNG), so the same
tile index gives the same tile on every machine."""
from __future__ import annotations

import numpy as np
import torch

TRAIN_SEED0 = 1000
HELDOUT_SEED0 = 900000


def make_tile(seed: int, size: int = 256, channels: int = 2):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = w = size
    v, u = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    a, b = rng.uniform(-8 * np.pi, 8 * np.pi, size=2) / size
    phi = a * u + b * v
    label = np.zeros((h, w), dtype=np.int64)
    for _ in range(int(rng.integers(1, 4))):
        x0, y0, x1, y1 = rng.uniform(0, size, size=4)
        step = rng.uniform(np.pi / 2, 2 * np.pi)
        dx, dy = x1 - x0, y1 - y0
        ln = max(np.hypot(dx, dy), 1e-6)
        side = ((u - x0) * dy - (v - y0) * dx) / ln          # signed distance to the infinite line
        t = ((u - x0) * dx + (v - y0) * dy) / (ln * ln)      # position along the segment
        phi = phi + step * (side > 0) * ((t >= 0) & (t <= 1))
        tc = np.clip(t, 0.0, 1.0)
        dist = np.hypot(u - (x0 + tc * dx), v - (y0 + tc * dy))
        label[dist <= 2.0] = 1
    phi = phi + 0.3 * rng.standard_normal((h, w))
    phi = np.angle(np.exp(1j * phi))
    if channels == 2:
        img = np.stack([np.cos(phi), np.sin(phi)], 0)
    else:
        img = (phi / np.pi)[None]
    return img.astype(np.float32), label


def make_tile_bowl(seed: int, size: int = 64, channels: int = 2):
    """Second synthetic task (the mIoU-parity experiment): one or two deformation BOWLS on a gentle ramp — inside a
    disc of radius r the phase gains A*(1 - d^2/r^2) (A = 3..8 pi, either sign: concentric fringes that are densest
    at the rim), label = 1 inside the disc. A region label (10-30 % positives) instead of make_tile's 2-px lines:
    U-Net-CA reaches a validation mIoU above 0.9 on it, where seed-to-seed scatter is small enough to resolve
    tenths of a point. Same PCG64-only seeding and (cos, sin) channels as make_tile."""
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    h = w = size
    v, u = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    a, b = rng.uniform(-4 * np.pi, 4 * np.pi, size=2) / size
    phi = a * u + b * v
    label = np.zeros((h, w), dtype=np.int64)
    for _ in range(int(rng.integers(1, 3))):
        cx, cy = rng.uniform(0.15 * size, 0.85 * size, size=2)
        r = rng.uniform(0.12 * size, 0.28 * size)
        amp = rng.uniform(3 * np.pi, 8 * np.pi) * (1.0 if rng.uniform() < 0.5 else -1.0)
        d2 = ((u - cx) ** 2 + (v - cy) ** 2) / (r * r)
        phi = phi + amp * np.maximum(0.0, 1.0 - d2)
        label[d2 <= 1.0] = 1
    phi = phi + 0.3 * rng.standard_normal((h, w))
    phi = np.angle(np.exp(1j * phi))
    if channels == 2:
        img = np.stack([np.cos(phi), np.sin(phi)], 0)
    else:
        img = (phi / np.pi)[None]
    return img.astype(np.float32), label


TASKS = {"lines": make_tile, "bowl": make_tile_bowl}


def make_batch(first_index: int, batch: int, size: int = 256, heldout: bool = False, channels: int = 2,
               task: str = "lines"):
    seed0 = HELDOUT_SEED0 if heldout else TRAIN_SEED0
    tiles = [TASKS[task](seed0 + first_index + i, size, channels) for i in range(batch)]
    x = torch.from_numpy(np.stack([t[0] for t in tiles], 0))
    y = torch.from_numpy(np.stack([t[1] for t in tiles], 0))
    return x, y


class SyntheticTiles(torch.utils.data.Dataset):
    """Dataset with the reference's (img [C,S,S] float32 in [-1,1], mask [S,S] int64) contract
    (VOCSegDataset.__getitem__, Unet-ChannalAttention.py:191-212) over the synthetic generator."""

    def __init__(self, count: int, size: int = 256, heldout: bool = False, channels: int = 2, offset: int = 0,
                 task: str = "lines"):
        self.count, self.size, self.heldout, self.channels, self.offset = count, size, heldout, channels, offset
        self.task = task

    def __len__(self):
        return self.count

    def __getitem__(self, idx: int):
        seed0 = HELDOUT_SEED0 if self.heldout else TRAIN_SEED0
        img, lab = TASKS[self.task](seed0 + self.offset + idx, self.size, self.channels)
        return torch.from_numpy(img), torch.from_numpy(lab)


# ---- the reference's on-disk format ----------------------------------------------
def reference_transforms(image_size: int):
    """The reference's `data_transforms` (Unet-ChannalAttention.py:428-432) without torchvision:
    Resize((S,S)) on a PIL image is PIL's bilinear `Image.resize` (torchvision calls exactly that for PIL
    inputs), ToTensor is uint8 -> float32 / 255 with a leading channel axis, Normalize(0.5, 0.5) is
    (x - 0.5) / 0.5. Returns a callable PIL 'L' image -> float32 tensor [1, S, S] in [-1, 1]."""
    from PIL import Image

    def apply(img):
        img = img.resize((image_size, image_size), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).astype(np.float32) / 255.0)
        return ((x - 0.5) / 0.5).unsqueeze(0)

    return apply


class VOCSegDataset(torch.utils.data.Dataset):
    """Reader for the reference's VOC-layout tile sets, same constructor and item contract as
    `VOCSegDataset` (Unet-ChannalAttention.py:167-212):

        voc_root/JPEGImages/<id>.jpg            grey-level tile          -> img  float32 [1, S, S]
        voc_root/SegmentationClass/<id>.png     mask, 255 = feature      -> mask int64 [S, S] in {0, 1}
        voc_root/ImageSets/Segmentation/<image_set>.txt   one id per line

    The mask goes through a nearest-neighbour resize, ToTensor (/255) and `.long()` in the reference
    (:201-210), i.e. floor(px / 255): 255 -> 1, every other grey level -> 0. `transforms` defaults to
    `reference_transforms(image_size)`; pass the reference's own torchvision Compose if torchvision is
    installed — anything mapping a PIL 'L' image to a tensor works."""

    def __init__(self, voc_root: str, image_size: int, image_set: str = "train", transforms=None):
        import os
        self.voc_root, self.image_size = voc_root, image_size
        self.transforms = transforms if transforms is not None else reference_transforms(image_size)
        self.image_dir = os.path.join(voc_root, "JPEGImages")
        self.mask_dir = os.path.join(voc_root, "SegmentationClass")
        self.image_set_path = os.path.join(voc_root, "ImageSets", "Segmentation", f"{image_set}.txt")
        if not os.path.exists(self.image_set_path):
            raise FileNotFoundError(f"ImageSets file not found: {self.image_set_path}")
        with open(self.image_set_path, "r") as f:
            self.ids = [line.strip() for line in f.readlines()]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, idx: int):
        import os
        from PIL import Image
        img_id = self.ids[idx]
        img = Image.open(os.path.join(self.image_dir, f"{img_id}.jpg")).convert("L")
        mask = Image.open(os.path.join(self.mask_dir, f"{img_id}.png")).convert("L")
        img = self.transforms(img)
        mask = mask.resize((self.image_size, self.image_size), Image.NEAREST)
        mask = torch.from_numpy((np.asarray(mask, dtype=np.uint8) // 255).astype(np.int64))
        return img, mask


class ShardedSampler(torch.utils.data.Sampler):
    """Data-parallel sampler: every rank draws the same seeded permutation of the dataset per
    epoch and keeps the indices `rank::world` of it, padded by wrap-around so that all ranks take the same
    number of steps (the gradient all-reduce needs that). `set_epoch(e)` reshuffles; numpy PCG64, no torch RNG."""

    def __init__(self, length: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.length, self.rank, self.world, self.shuffle, self.seed = length, rank, world, shuffle, seed
        self.epoch = 0
        self.per_rank = -(-length // world)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self):
        return self.per_rank

    def __iter__(self):
        if self.shuffle:
            order = np.random.Generator(np.random.PCG64([self.seed, self.epoch])).permutation(self.length)
        else:
            order = np.arange(self.length)
        total = self.per_rank * self.world
        if total > self.length:
            order = np.resize(order, total)            # cyclic wrap-around padding
        return iter(order[self.rank:total:self.world].tolist())


def make_loader(dataset, batch_size: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0,
                num_workers: int = 0, drop_last: bool = False):
    """DataLoader as the reference builds it (:436-451: pinned memory, worker processes) with the rank's
    shard of the data; `.to(device, non_blocking=True)` in the training loop then overlaps the H2D copy."""
    sampler = ShardedSampler(len(dataset), rank, world, shuffle, seed)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                                       pin_memory=torch.cuda.is_available(), drop_last=drop_last,
                                       persistent_workers=num_workers > 0)


class SeededBatches:
    """A DataLoader stand-in with the two things the reference's train_model / validate_model use (iteration over
    (images, masks) batches and `.dataset` for len(), Unet-ChannalAttention.py:338,359): pre-built batches served in a
    PCG64-seeded order that is reshuffled on every pass (shuffle=True, as the reference's train loader) or in index
    order (validation). Used by BOTH sides of the mIoU-parity experiment so that they see identical tiles in
    identical order (torch's own RandomSampler would tie the order to torch's RNG state)."""

    def __init__(self, batches, shuffle: bool, seed: int = 0, device=None):
        self.batches = [(x.to(device), y.to(device)) for x, y in batches] if device is not None else list(batches)
        self.shuffle = shuffle
        self.rng = np.random.Generator(np.random.PCG64([seed, 4242]))
        self.dataset = range(sum(int(x.shape[0]) for x, _ in self.batches))

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        order = self.rng.permutation(len(self.batches)) if self.shuffle else np.arange(len(self.batches))
        for i in order:
            yield self.batches[int(i)]


class DevicePrefetcher:
    """Feeds a training loop from a host loader without stalling the step: the reference copies every
    batch inside the step (`imgs.to(device)`, Unet-ChannalAttention.py:339-340, from the pinned-memory DataLoader of
    :436-451); here batch i+1 is copied on a COPY STREAM (pinned source, non_blocking) while step i computes, into one of
    two device slots, and the compute stream only waits for the copy's event. Iterating yields (images, masks) device
    tensors. A slot is refilled only after the step that used it has been enqueued (`record_stream`-free: the consumer's
    position on the compute stream is recorded as an event when the NEXT batch is requested and the copy stream waits for
    it), so the copy of batch i+2 never overwrites tensors that step i still reads.

    `loader`: any iterable of (images, masks) CPU tensors (make_loader(...), SeededBatches, a list). Tensors that are not
    pinned are staged through this object's own pinned buffers.
    `compact_masks`: int64 masks whose values all lie in 0..255 (class indices and the ignore value 255 of the reference's
    VOC layout) cross the host link as uint8 — an eighth of their bytes, 9.4 instead of 16.8 MB per batch of config 2 —
    and are widened to int64 into the device slot on the copy stream; the yielded tensors are the same int64 masks bit for
    bit. A batch with any other value takes the plain path."""

    def __init__(self, loader, device, slots: int = 2, compact_masks: bool = True):
        self.loader, self.device = loader, torch.device(device)
        self.compact_masks = compact_masks
        self._pin_u8 = [None] * max(2, slots)
        self._dev_u8 = [None] * max(2, slots)
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher needs a ROCm device (there is no CPU fallback on the HIP path)")
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.nslots = max(2, slots)
        self._dev = [None] * self.nslots
        self._pin = [None] * self.nslots
        self._free = [None] * self.nslots          # event: the step that last used this slot has been enqueued
        self._ready = [None] * self.nslots         # event: the last copy into this slot (out of its pinned buffer) is done
        self.dataset = getattr(loader, "dataset", None)

    def __len__(self):
        return len(self.loader)

    def _stage(self, slot: int, batch):
        x, y = batch
        if self._dev[slot] is None or self._dev[slot][0].shape != x.shape or self._dev[slot][1].shape != y.shape:
            self._dev[slot] = (torch.empty(x.shape, dtype=x.dtype, device=self.device),
                               torch.empty(y.shape, dtype=y.dtype, device=self.device))
            self._pin[slot] = None
            # the caching allocator may hand back a block whose last use on the compute stream is still queued (first use
            # of a slot, or a re-allocation for the last, smaller batch): the first copy into it waits for that stream
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        compact = False
        if self.compact_masks and y.dtype == torch.int64 and y.device.type == "cpu" and y.numel() > 0:
            # numpy, not torch: a torch reduction wakes the whole intra-op thread pool (128 threads on a 16-core GPU box:
            # 23 ms per call inside a training loop, against 0.17 ms for these two single-threaded passes over 1 M values)
            yn = y.numpy()
            compact = int(yn.min()) >= 0 and int(yn.max()) <= 255
        if compact:
            if self._pin_u8[slot] is None or self._pin_u8[slot].shape != y.shape:
                self._pin_u8[slot] = torch.empty(y.shape, dtype=torch.uint8).pin_memory()
                self._dev_u8[slot] = torch.empty(y.shape, dtype=torch.uint8, device=self.device)
                self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            if self._ready[slot] is not None:
                self._ready[slot].synchronize()       # the previous copy out of this pinned buffer is done
            np.copyto(self._pin_u8[slot].numpy(), yn, casting="unsafe")     # int64 -> uint8 on the host, exact for 0..255
        src = []
        for k, t in enumerate((x, y)):
            if k == 1 and compact:
                src.append(None)
                continue
            if not t.is_pinned():
                if self._pin[slot] is None:
                    self._pin[slot] = [torch.empty(x.shape, dtype=x.dtype).pin_memory(), torch.empty(y.shape, dtype=y.dtype).pin_memory()]
                if self._ready[slot] is not None:
                    self._ready[slot].synchronize()   # the previous copy OUT of this pinned buffer (two batches ago) is done
                self._pin[slot][k].copy_(t)
                t = self._pin[slot][k]
            src.append(t)
        with torch.cuda.stream(self.copy_stream):
            if self._free[slot] is not None:
                self.copy_stream.wait_event(self._free[slot])
            self._dev[slot][0].copy_(src[0], non_blocking=True)
            if compact:
                self._dev_u8[slot].copy_(self._pin_u8[slot], non_blocking=True)
                self._dev[slot][1].copy_(self._dev_u8[slot])          # widened on the device, on the copy stream
            else:
                self._dev[slot][1].copy_(src[1], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._ready[slot] = ready
        return ready

    def __iter__(self):
        it = iter(self.loader)
        pending = []                                 # (slot, ready event)
        slot = 0
        try:
            pending.append((slot, self._stage(slot, next(it))))
        except StopIteration:
            return
        while pending:
            cur, ready = pending.pop(0)
            nxt = (cur + 1) % self.nslots
            try:
                batch = next(it)
            except StopIteration:
                batch = None
            if batch is not None:
                pending.append((nxt, self._stage(nxt, batch)))
            torch.cuda.current_stream(self.device).wait_event(ready)
            yield self._dev[cur]
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))       # the consumer's launches are enqueued up to here
            self._free[cur] = done
