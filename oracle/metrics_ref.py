"""CPU restatement of the scoring metrics (TEST INFRASTRUCTURE ONLY -- never imported by the product).

SSIM follows skimage.metrics.structural_similarity as code/GAN/psnr_ssim_metric.py:91-92 calls it
(`structural_similarity(t2, t2_gen, data_range=256)`, every other argument at its default):
win_size 7, uniform filter, K1 = 0.01, K2 = 0.03, use_sample_covariance=True, float64 arithmetic,
mean over the image cropped by (win_size - 1) // 2 on every axis.  scikit-image is pinned at 0.18.1 by
the reference (REQUIREMENTS.txt) and is NOT installed here, so this is a restatement of its published
algorithm on scipy.ndimage.uniform_filter (the same primitive skimage uses): **parity unpinned**
against skimage itself.
"""
import numpy as np
from scipy.ndimage import uniform_filter


def structural_similarity(im1, im2, data_range=256.0, win_size=7, K1=0.01, K2=0.03):
    im1 = np.asarray(im1, dtype=np.float64)
    im2 = np.asarray(im2, dtype=np.float64)
    ndim = im1.ndim
    NP = win_size ** ndim
    cov_norm = NP / (NP - 1)                       # sample covariance
    ux = uniform_filter(im1, size=win_size)
    uy = uniform_filter(im2, size=win_size)
    uxx = uniform_filter(im1 * im1, size=win_size)
    uyy = uniform_filter(im2 * im2, size=win_size)
    uxy = uniform_filter(im1 * im2, size=win_size)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1 = (K1 * data_range) ** 2
    C2 = (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    crop = tuple(slice(pad, s - pad) for s in S.shape)
    return float(S[crop].mean(dtype=np.float64))


def scale_intensity_range_percentiles(img, lower=1.0, upper=99.0, b_min=-1.0, b_max=1.0, clip=True):
    """MONAI 0.4.0 `ScaleIntensityRangePercentiles.__call__` with relative=False followed by
    `ScaleIntensityRange.__call__` (3rd-party source restated; call site code/GAN/GAN_final.py:386-394)."""
    img = np.asarray(img)
    a_min = np.percentile(img, lower)
    a_max = np.percentile(img, upper)
    if a_max - a_min == 0.0:
        return img - a_min
    out = (img - a_min) / (a_max - a_min)
    out = out * (b_max - b_min) + b_min
    if clip:
        out = np.clip(out, b_min, b_max)
    return out
