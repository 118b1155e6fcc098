#!/usr/bin/env python3
"""Is the 4-7 us a K-stepped launch spends before its first MFMA instruction fetch, data fetch, or neither?
(development aid; stamps build).  The same kernel is launched N times back to back (a) on the SAME operands,
(b) on FRESH operands every time (different input, weights, output buffers: cold data, warm code if the instruction
cache survives a launch), (c) alternating with a different big kernel in between (cold code).  Prints the median
prologue / K loop / epilogue of every launch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MPGAN_LIB_PATH", os.path.join(ROOT, "cross-modality-minipig-gan_amd", "libmpgan_hip_stamps.so"))
import numpy as np
import torch
from mpgan_amd import ops
from mpgan_amd._lib import lib

SLOTS, CAP = 12, 1024


def main():
    L = lib()
    tick = 1e3 / L.mpgan_debug_clock_khz()
    n, hw, cin, cout = 16, 32, 64, 128
    g = ops.ConvGeom(n, (1, hw, hw), cin, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    g2 = ops.ConvGeom(n, (1, 64, 64), 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1))          # the "other" kernel (patch form)
    N = 6
    xs = [torch.rand(n, 1, hw, hw, cin, device="cuda") for _ in range(N)]
    ws = [ops.pack_weight((torch.rand(cout, cin, 3, 3, device="cuda") - 0.5) / 24) for _ in range(N)]
    ys = [torch.empty(n, 1, hw, hw, cout, device="cuda") for _ in range(N)]
    x2 = torch.rand(n, 1, 64, 64, 32, device="cuda")
    w2 = ops.pack_weight((torch.rand(32, 32, 3, 3, device="cuda") - 0.5) / 17)
    y2 = torch.empty(n, 1, 64, 64, 32, device="cuda")
    flush = torch.empty(96 * 1024 * 1024, device="cuda")                               # 384 MB: evicts L2 and MALL

    def run(mode):
        buf = torch.zeros(2 * N * CAP * SLOTS, dtype=torch.int64, device="cuda")
        flush.fill_(1.0)
        torch.cuda.synchronize()
        assert L.mpgan_debug_stamps(buf.data_ptr(), 2 * N, CAP) == 0
        idx = []
        for i in range(N):
            k = 0 if mode == "same" else i
            ops.conv_forward(g, xs[k], ws[k], None, ys[k])
            idx.append(L.mpgan_debug_stamps_used() - 1)
            if mode == "alternate":
                ops.conv_forward(g2, x2, w2, None, y2)
        torch.cuda.synchronize()
        L.mpgan_debug_stamps(None, 0, 0)
        s = buf.cpu().numpy().reshape(2 * N, CAP, SLOTS).astype(np.float64)
        out = []
        for j in idx:
            b = s[j][s[j][:, 0] > 0]
            med = lambda v: float(np.median(v)) * tick
            out.append((med(b[:, 1] - b[:, 0]), med(b[:, 2] - b[:, 1]), med(b[:, 7] - b[:, 3]), (b[:, 7].max() - b[:, 0].min()) * tick))
        print(f"{mode:10s} " + "  ".join(f"[pro {p:4.1f} K {k:5.1f} epi {e:4.1f} span {sp:5.1f}]" for p, k, e, sp in out))

    for _ in range(2):
        ops.conv_forward(g, xs[0], ws[0], None, ys[0])
        ops.conv_forward(g2, x2, w2, None, y2)
    torch.cuda.synchronize()
    print(f"64->128 k3 @32^2 bs16 (K-stepped, in-block split-K), {N} launches each; microseconds")
    for mode in ("same", "fresh", "alternate", "same"):
        run(mode)


if __name__ == "__main__":
    main()
