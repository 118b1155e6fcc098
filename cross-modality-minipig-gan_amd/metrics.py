"""On-device evaluation metrics of the reference's offline scripts (SURVEY.md section 8(f)
row N2): 0..255 rescale + round as code/GAN/inferrence.py:188-204 applies it, then MAE / MSE /
PSNR and SSIM (data_range 256, code/GAN/psnr_ssim_metric.py:88-106; code/GAN/metrics.py:213-223).
SSIM restates skimage.metrics.structural_similarity's published algorithm (skimage itself is not in
this image: parity is pinned on a scipy.ndimage restatement in oracle/metrics_ref.py only)."""
from __future__ import annotations

import ctypes as C
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
    """The inference script's scoring: both volumes rescaled to 0..255 and rounded, then compared
    (MAE / MSE / PSNR; plus SSIM when the tensors are (H, W) or (D, H, W))."""
    g, t = rescale_0_255(generated), rescale_0_255(ground_truth)
    out = image_errors(g, t, 256.0)
    if g.dim() in (2, 3) and min(g.shape[-2:]) >= 7 and (g.dim() == 2 or g.shape[0] >= 7):
        out["ssim"] = ssim(g, t, 256.0)
    return out


def ssim(a: torch.Tensor, b: torch.Tensor, data_range: float = 256.0) -> torch.Tensor:
    """structural_similarity(a, b, data_range=256) of two (H, W) slices or (D, H, W) volumes
    (code/GAN/psnr_ssim_metric.py:91-92): 7-wide uniform window, K1 = 0.01, K2 = 0.03, sample
    covariance, mean over the interior.  Returns a device scalar."""
    if a.shape != b.shape or a.dim() not in (2, 3):
        raise ValueError("ssim: expects two same-shape (H, W) or (D, H, W) tensors")
    a, b = a.contiguous().float(), b.contiguous().float()
    dhw = (C.c_int32 * 3)(*((1,) * (3 - a.dim()) + tuple(a.shape)))
    need = int(lib().mpgan_ssim_workspace(dhw))
    if need < 0:
        raise ValueError(f"ssim: extents {tuple(a.shape)} are below the 7-wide window")
    ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=a.device)
    out = torch.empty(1, device=a.device)
    check(lib().mpgan_ssim(a.data_ptr(), b.data_ptr(), dhw, float(data_range), ws.data_ptr(), ws.numel() * 8,
                           out.data_ptr(), _stream()), "ssim")
    return out[0]
