"""On-device evaluation metrics of the reference's offline scripts (SURVEY.md section 8(f)
row N2): 0..255 rescale + round as code/GAN/inferrence.py:188-204 applies it, then MAE / MSE /
PSNR (data_range 256, code/GAN/psnr_ssim_metric.py:88-106; code/GAN/metrics.py:213-223).
SSIM (skimage) is not built."""
from __future__ import annotations

from typing import Dict

import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def rescale_0_255(x: torch.Tensor, do_round: bool = True) -> torch.Tensor:
    """ScaleIntensityRangePercentiles(lower=0, upper=100, b_min=0, b_max=255, clip=True) + round."""
    x = x.contiguous()
    part = torch.empty(int(lib().mpgan_metric_partials()), device=x.device)
    mm = torch.empty(2, device=x.device)
    y = torch.empty_like(x)
    check(lib().mpgan_rescale_minmax(x.data_ptr(), x.numel(), 0.0, 255.0, int(do_round), part.data_ptr(),
                                     mm.data_ptr(), y.data_ptr(), _stream()), "rescale_minmax")
    return y


def image_errors(a: torch.Tensor, b: torch.Tensor, data_range: float = 256.0) -> Dict[str, torch.Tensor]:
    """MAE, MSE and PSNR between two same-shape device tensors (scalars stay on the device)."""
    if a.shape != b.shape:
        raise ValueError("image_errors: shape mismatch")
    a, b = a.contiguous(), b.contiguous()
    part = torch.empty(int(lib().mpgan_metric_partials()), device=a.device)
    out = torch.empty(3, device=a.device)
    check(lib().mpgan_image_errors(a.data_ptr(), b.data_ptr(), a.numel(), float(data_range), part.data_ptr(),
                                   out.data_ptr(), _stream()), "image_errors")
    return {"mae": out[0], "mse": out[1], "psnr": out[2]}


def score_volume(generated: torch.Tensor, ground_truth: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The inference script's scoring: both volumes rescaled to 0..255 and rounded, then compared."""
    return image_errors(rescale_0_255(generated), rescale_0_255(ground_truth), 256.0)
