"""insar_unet_ca_amd — MI355X-native U-Net-CA training hot path (drop-in for
Createroner/InSAR-Unet-CA's Unet-ChannalAttention.py model/loss/optimizer entry points)."""
from ._lib import InsarError, LIB_PATH  # noqa: F401
from .data import DevicePrefetcher, ShardedSampler, SyntheticTiles, VOCSegDataset, make_loader, reference_transforms  # noqa: F401
from .loss import CrossEntropyLoss, DiceCELoss, DiceLoss  # noqa: F401
from .modules import ChannelAttentionModule, DoubleConv, MaxPool2d, SELayer, UNet  # noqa: F401
from .deeplab import DeepLabV3_SingleChannel_Attn  # noqa: F401
from .optim import Adam  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401
from .train import compute_metrics, save_history, train_model, validate_model  # noqa: F401

__all__ = ["UNet", "DeepLabV3_SingleChannel_Attn", "DoubleConv", "SELayer", "ChannelAttentionModule", "MaxPool2d", "CrossEntropyLoss", "DiceLoss", "DiceCELoss", "Adam", "GraphedTrainStep",
           "compute_metrics", "train_model", "validate_model", "save_history", "VOCSegDataset", "SyntheticTiles",
           "ShardedSampler", "DevicePrefetcher", "make_loader", "reference_transforms", "InsarError", "LIB_PATH"]
