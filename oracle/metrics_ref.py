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
