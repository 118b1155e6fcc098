"""Intensity pre-processing on the device: the array half of SURVEY.md section 8(f) row N3.

`ScaleIntensityRangePercentilesd(keys, lower=1, upper=99, b_min=-1, b_max=1, clip=True)` is the last
transform before the training path (code/GAN/GAN_final.py:386-394; MONAI 0.4.0:
`a_min, a_max = np.percentile(img, lower), np.percentile(img, upper)` then `ScaleIntensityRange`).
`resample_to_identity_grid` is the array half of the transform before it, `ResampleT1T2d`
(code/GAN/transforms.py:79-213): linear interpolation onto the 256 mm identity-direction grid.  Reading the
NIfTI files themselves (ITK) stays out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def percentiles(x: torch.Tensor, q: Sequence[float]) -> torch.Tensor:
    """np.percentile(x, q) (linear interpolation) for one or two percentiles; returns a device tensor."""
    q = [float(v) for v in q]
    if not 1 <= len(q) <= 2:
        raise ValueError("percentiles: one or two percentiles per call")
    x = x.contiguous().float()
    need = int(lib().mpgan_percentile_workspace())
    ws = torch.empty((need + 7) // 8, dtype=torch.int64, device=x.device)
    out = torch.empty(len(q), device=x.device)
    qh = (C.c_double * len(q))(*q)
    check(lib().mpgan_percentiles(x.data_ptr(), x.numel(), qh, len(q), ws.data_ptr(), ws.numel() * 8, out.data_ptr(),
                                  _stream()), "percentiles")
    return out


def scale_intensity_range_percentiles(x: torch.Tensor, lower: float = 1.0, upper: float = 99.0, b_min: float = -1.0,
                                      b_max: float = 1.0, clip: bool = True) -> torch.Tensor:
    """MONAI 0.4.0 ScaleIntensityRangePercentiles(lower, upper, b_min, b_max, clip, relative=False) on
    one image/volume (any shape: the percentiles are taken over all of its elements)."""
    x = x.contiguous().float()
    mm = percentiles(x, (lower, upper))
    y = torch.empty_like(x)
    check(lib().mpgan_scale_intensity_range(x.data_ptr(), x.numel(), mm.data_ptr(), float(b_min), float(b_max),
                                            int(clip), y.data_ptr(), _stream()), "scale_intensity_range")
    return y


def resample_to_identity_grid(vol: torch.Tensor, origin, spacing, direction, output_size=(128, 128, 128),
                              extent_mm: float = 256.0) -> torch.Tensor:
    """`ResampleT1T2d` (code/GAN/transforms.py:79-213) for one volume: `vol` is the (D,H,W) array of an ITK image
    with `origin` / `spacing` in ITK's (x,y,z) order and a 3x3 `direction`; returns the (D,H,W) = reversed
    `output_size` (x,y,z) resampled array on the reference grid origin = -size/2, spacing = extent_mm/size,
    identity direction (identity transform, linear interpolation, 0 outside the input)."""
    if vol.dim() != 3:
        raise ValueError("resample_to_identity_grid: expected a (D,H,W) volume")
    vol = vol.contiguous().float()
    out_dhw = tuple(int(v) for v in reversed(tuple(output_size)))
    out = torch.empty(out_dhw, device=vol.device)
    i3 = lambda v: (C.c_int32 * 3)(*[int(x) for x in v])
    d3 = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
    dirs = [float(x) for row in direction for x in row] if len(direction) == 3 else [float(x) for x in direction]
    check(lib().mpgan_resample_to_identity_grid(vol.data_ptr(), i3(vol.shape), d3(origin), d3(spacing), d3(dirs),
                                                i3(out_dhw), float(extent_mm), out.data_ptr(), _stream()),
          "resample_to_identity_grid")
    return out
