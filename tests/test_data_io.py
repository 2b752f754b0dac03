"""The data formats either side of the hot path: the reference's VOC-layout tile
reader, the data-parallel sampler and the metrics-history file. CPU only. torchvision is not installed
here, so the reader is checked against the PIL / numpy arithmetic the reference's transform chain reduces
to (Unet-ChannalAttention.py:191-212, 428-432), stated independently below."""
import json
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from insar_unet_ca_amd.data import ShardedSampler, VOCSegDataset, make_loader, reference_transforms  # noqa: E402
from insar_unet_ca_amd.train import save_history  # noqa: E402


def _make_voc(root, ids, size=40, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    for d in ("JPEGImages", "SegmentationClass", os.path.join("ImageSets", "Segmentation")):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for i in ids:
        img = (rng.random((size, size)) * 255).astype(np.uint8)
        Image.fromarray(img, "L").save(os.path.join(root, "JPEGImages", f"{i}.jpg"), quality=95)
        mask = rng.choice(np.array([0, 128, 254, 255], dtype=np.uint8), size=(size, size), p=[0.6, 0.1, 0.1, 0.2])
        Image.fromarray(mask, "L").save(os.path.join(root, "SegmentationClass", f"{i}.png"))
    with open(os.path.join(root, "ImageSets", "Segmentation", "train.txt"), "w") as f:
        f.write("\n".join(ids[:-1]) + "\n")
    with open(os.path.join(root, "ImageSets", "Segmentation", "val.txt"), "w") as f:
        f.write(ids[-1] + "\n")


def test_voc_reader_item_contract(tmp_path):
    ids = ["tile_000", "tile_001", "tile_002", "tile_003"]
    _make_voc(str(tmp_path), ids)
    S = 32
    ds = VOCSegDataset(str(tmp_path), image_size=S, image_set="train")
    assert len(ds) == 3 and len(VOCSegDataset(str(tmp_path), S, "val")) == 1
    img, mask = ds[1]
    assert img.dtype == torch.float32 and img.shape == (1, S, S) and float(img.min()) >= -1.0 and float(img.max()) <= 1.0
    assert mask.dtype == torch.int64 and mask.shape == (S, S) and set(mask.unique().tolist()) <= {0, 1}
    # independent statement of the reference chain
    pil = Image.open(os.path.join(str(tmp_path), "JPEGImages", "tile_001.jpg")).convert("L").resize((S, S), Image.BILINEAR)
    expect = (np.asarray(pil, dtype=np.float32) / 255.0 - 0.5) / 0.5
    np.testing.assert_array_equal(img[0].numpy(), expect)
    pm = Image.open(os.path.join(str(tmp_path), "SegmentationClass", "tile_001.png")).convert("L").resize((S, S), Image.NEAREST)
    np.testing.assert_array_equal(mask.numpy(), (np.asarray(pm) == 255).astype(np.int64))     # 128 and 254 -> 0
    # a caller-supplied transform (e.g. the reference's torchvision Compose) is used as is
    ds2 = VOCSegDataset(str(tmp_path), S, "train", transforms=lambda im: torch.zeros(1, S, S))
    assert float(ds2[0][0].abs().max()) == 0.0
    with pytest.raises(FileNotFoundError):
        VOCSegDataset(str(tmp_path), S, "test")


def test_reference_transforms_range():
    t = reference_transforms(16)
    white = t(Image.fromarray(np.full((20, 20), 255, np.uint8), "L"))
    black = t(Image.fromarray(np.zeros((20, 20), np.uint8), "L"))
    assert torch.all(white == 1.0) and torch.all(black == -1.0)


@pytest.mark.parametrize("length,world", [(10, 1), (10, 2), (11, 4), (3, 8)])
def test_sharded_sampler_partitions(length, world):
    shards = [list(ShardedSampler(length, r, world, shuffle=True, seed=5)) for r in range(world)]
    per = -(-length // world)
    assert all(len(s) == per for s in shards)                       # every rank takes the same number of steps
    seen = [i for s in shards for i in s]
    assert set(seen) == set(range(length))                          # every tile is visited
    assert len(seen) - length == per * world - length               # only the wrap-around padding repeats
    a = ShardedSampler(length, 0, world, True, seed=5)
    b = ShardedSampler(length, 0, world, True, seed=5)
    b.set_epoch(1)
    if length > 3:
        assert list(a) != list(b)                                   # reshuffled per epoch
    assert list(ShardedSampler(length, 0, world, shuffle=False)) == np.resize(np.arange(length), per * world)[0::world].tolist()
    with pytest.raises(ValueError):
        ShardedSampler(4, 2, 2)


def test_loader_over_voc_reader(tmp_path):
    ids = [f"t{i}" for i in range(7)]
    _make_voc(str(tmp_path), ids)
    ds = VOCSegDataset(str(tmp_path), 16, "train")
    batches = [list(make_loader(ds, batch_size=2, rank=r, world=2, shuffle=False)) for r in range(2)]
    assert [len(b) for b in batches] == [2, 2]
    x, y = batches[0][0]
    assert x.shape == (2, 1, 16, 16) and y.shape == (2, 16, 16) and y.dtype == torch.int64


def test_history_file_schema(tmp_path):
    hist = [{"epoch": 1, "train_loss": torch.tensor(0.7), "train_miou": 0.4, "val_loss": 0.6, "val_miou": torch.tensor(0.5)}]
    path = os.path.join(str(tmp_path), "training_metrics", "history.json")
    save_history(hist, path)
    back = json.load(open(path))
    assert back == [{"epoch": 1, "train_loss": pytest.approx(0.7), "train_miou": 0.4, "val_loss": 0.6, "val_miou": 0.5}]


def test_checkpoint_interchange_with_the_reference(tmp_path):
    """`.pth` files written by `torch.save(model.state_dict())` (the reference's checkpoint, :382-387) load
    into the other implementation with strict=True, in both directions, key order included. In-container
    only: the reference tree is not present on the GPU box."""
    from oracle import ref_loader
    if not ref_loader.reference_available():
        pytest.skip("reference tree not mounted")
    import insar_unet_ca_amd as iu
    ref = ref_loader.load_reference_unet_ca()
    theirs = ref.UNet(in_channels=1, num_classes=2, use_se=True)
    ours = iu.UNet(in_channels=1, num_classes=2, use_se=True)
    assert list(theirs.state_dict().keys()) == list(ours.state_dict().keys())
    p1 = os.path.join(str(tmp_path), "theirs.pth")
    torch.save(theirs.state_dict(), p1)
    ours.load_state_dict(torch.load(p1), strict=True)
    for (k, a), (_, b) in zip(theirs.state_dict().items(), ours.state_dict().items()):
        assert a.dtype == b.dtype and torch.equal(a, b), k
    p2 = os.path.join(str(tmp_path), "ours.pth")
    torch.save(ours.state_dict(), p2)
    theirs.load_state_dict(torch.load(p2), strict=True)
    # Unet.py-equivalent (use_se=False) keeps the same contract minus the SE entries
    assert list(ref.UNet(2, 2, False).state_dict().keys()) == list(iu.UNet(2, 2, False).state_dict().keys())
