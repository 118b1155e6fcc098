"""TEST INFRASTRUCTURE: numpy restatement of `ResampleT1T2d` (code/GAN/transforms.py:79-213) =
itk.resample_image_filter(identity transform, LinearInterpolateImageFunction, reference image) as ITK 5 defines
it (ITK is not installable in this image, so this restatement -- like the kernel -- is PARITY UNPINNED against
ITK itself; it pins the kernel against an independent, vectorised implementation of the published algorithm:
ImageFunction::IsInsideBuffer's [-0.5, size-0.5) test, LinearInterpolateImageFunction's clamped base index and
dropped out-of-range neighbour, default pixel 0)."""
import numpy as np


def resample_to_identity_grid(vol, origin, spacing, direction, output_size=(128, 128, 128), extent_mm=256.0):
    """vol: (D,H,W) array of an ITK image; origin / spacing in (x,y,z); direction 3x3; output_size (x,y,z)."""
    vol = np.asarray(vol, dtype=np.float64)
    size_in = np.array(vol.shape[::-1])                       # (x, y, z)
    out_size = np.array(output_size)
    origin_out = -out_size / 2.0                              # transforms.py:144
    spacing_out = extent_mm / out_size                        # transforms.py:145
    A = np.asarray(direction, dtype=np.float64).reshape(3, 3) @ np.diag(np.asarray(spacing, dtype=np.float64))
    M = np.linalg.inv(A)
    ix, iy, iz = np.meshgrid(np.arange(out_size[0]), np.arange(out_size[1]), np.arange(out_size[2]), indexing="ij")
    p = np.stack([origin_out[0] + ix * spacing_out[0], origin_out[1] + iy * spacing_out[1],
                  origin_out[2] + iz * spacing_out[2]], axis=-1)
    c = (p - np.asarray(origin, dtype=np.float64)) @ M.T      # continuous index (x, y, z)
    inside = np.all((c >= -0.5) & (c < size_in - 0.5), axis=-1)
    base = np.clip(np.floor(c).astype(np.int64), 0, None)
    frac = np.clip(c - base, 0.0, None)
    frac = np.where(base + 1 > size_in - 1, 0.0, frac)
    base = np.minimum(base, size_in - 1)
    nxt = np.minimum(base + (frac > 0), size_in - 1)

    def at(bx, by, bz):
        return vol[bz, by, bx]

    x, y, z = frac[..., 0], frac[..., 1], frac[..., 2]
    v000, v100 = at(base[..., 0], base[..., 1], base[..., 2]), at(nxt[..., 0], base[..., 1], base[..., 2])
    v010, v110 = at(base[..., 0], nxt[..., 1], base[..., 2]), at(nxt[..., 0], nxt[..., 1], base[..., 2])
    v001, v101 = at(base[..., 0], base[..., 1], nxt[..., 2]), at(nxt[..., 0], base[..., 1], nxt[..., 2])
    v011, v111 = at(base[..., 0], nxt[..., 1], nxt[..., 2]), at(nxt[..., 0], nxt[..., 1], nxt[..., 2])
    x00, x10 = v000 + x * (v100 - v000), v010 + x * (v110 - v010)
    x01, x11 = v001 + x * (v101 - v001), v011 + x * (v111 - v011)
    y0, y1 = x00 + y * (x10 - x00), x01 + y * (x11 - x01)
    out = np.where(inside, y0 + z * (y1 - y0), 0.0)
    return np.ascontiguousarray(out.transpose(2, 1, 0)).astype(np.float32)      # (D,H,W)
