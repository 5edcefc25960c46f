"""gencomm_amd -- MI355X (gfx950) implementation of GenComm's generative-communication hot path.

Public surface mirrors the reference's plugin API:

    GenComm(model_cfg)                 opencood/models/gencomm_modules/cond_diff.py:185
    DiffusionUNet(config)              opencood/models/gencomm_modules/unet.py:198
    Enhancer(C, win_size, num_heads)   opencood/models/gencomm_modules/enhancer.py:359
    AttFusion(feature_dims)            opencood/models/fuse_modules/fusion_in_one.py:126
    regroup, normalize_pairwise_tfm    fusion_in_one.py:48, opencood/utils/transformation_utils.py:68
    MessageExtractorv2(in_ch, out_ch)  opencood/models/gencomm_modules/message_extractor_v2.py:109

All compute runs in hand-written HIP kernels behind the C ABI of ``include/gencomm_hip.h``.
"""
from .cond_diff import GenComm
from .enhancer import Enhancer
from .fusion import AttFusion, normalize_pairwise_tfm, regroup
from .message_extractor import MessageExtractorv2
from .unet import DiffusionUNet



def set_denoise_dtype(dtype) -> None:
    """Arithmetic of the denoise loop, process-wide (``gencomm_set_mode(GENCOMM_MODE_ARITH, ...)``):
    ``torch.float32`` (default: fp32 tensors, fp32-grade products on the f16 matrix pipe), ``"exact_fp32"`` (exact-fp32 MFMA
    kernels) or ``torch.bfloat16`` (bf16 storage of the UNet's 8-channel maps and single bf16 products -- the analogue of the
    reference's ``--half`` / autocast runs, ``train_ddp.py:139-141``; inference only)."""
    import torch
    from . import _lib
    value = {torch.float32: 0, "float32": 0, "exact_fp32": 1, torch.bfloat16: 2, "bfloat16": 2, "bf16": 2}[dtype]
    _lib.check(_lib.lib().gencomm_set_mode(_lib.MODE_ARITH, value), "gencomm_set_mode")


__all__ = ["GenComm", "DiffusionUNet", "Enhancer", "AttFusion", "MessageExtractorv2", "regroup", "normalize_pairwise_tfm",
           "set_denoise_dtype"]
