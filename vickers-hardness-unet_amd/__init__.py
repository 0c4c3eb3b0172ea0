"""vickers-hardness-unet_amd — MI355X (gfx950) implementation of the one compute path of
ZooMEISTER/vickers-hardness-Unet: the ResNet-34 U-Net forward / BCE+Dice / backward / AdamW step.

Import by string (the directory name carries hyphens)::

    import importlib
    vk = importlib.import_module("vickers-hardness-unet_amd")
    model = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).cuda()
    loss_dice = vk.losses.DiceLoss(mode="binary")
    optimizer = vk.adamw_for(model, lr=5e-5, weight_decay=1e-4)

Everything heavy runs in libvkunet.so (hand-written HIP); there is no CPU fallback."""
from . import _lib, augment, geometry, losses, parallel, prepost, segmetrics, synthetic  # noqa: F401
from .augment import AugmentSampler, DeviceDataset  # noqa: F401
from .geometry import (postprocess_minarearect_batch, postprocess_minarearect_multi, postprocess_quadrilateral_batch,  # noqa: F401
                       postprocess_quadrilateral_multi)
from ._lib import VkError, build, lib  # noqa: F401
from .losses import BCEDiceLoss, DiceLoss  # noqa: F401
from .segmetrics import dice_coef, iou_coef, seg_metrics, seg_metrics_device  # noqa: F401
from .optim import FusedAdamW, GradScaler, adamw_for  # noqa: F401
from .parallel import GradientReducer, all_reduce_scalars, broadcast_model, make_data_parallel  # noqa: F401
from .prepost import Segmenter, predict_mask  # noqa: F401
from .synthetic import seed_everything, synthetic_batch  # noqa: F401
from .unet import Unet, build_model  # noqa: F401

__all__ = ["Unet", "build_model", "DiceLoss", "BCEDiceLoss", "FusedAdamW", "GradScaler", "adamw_for", "GradientReducer",
           "make_data_parallel", "broadcast_model", "all_reduce_scalars", "dice_coef", "iou_coef", "seg_metrics", "seg_metrics_device", "synthetic_batch", "seed_everything",
           "VkError", "build", "lib", "losses", "segmetrics", "synthetic", "parallel", "prepost", "Segmenter", "predict_mask", "geometry", "augment", "AugmentSampler", "DeviceDataset",
           "postprocess_minarearect_multi", "postprocess_minarearect_batch", "postprocess_quadrilateral_multi", "postprocess_quadrilateral_batch"]
