#!/usr/bin/env python3
"""Eval-mode generator inference (code/GAN/inferrence.py:97-110,169-170):
BatchNorm from running statistics, no_grad.  Development aid."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.networks import CasNetGenerator


def run(shape, dims, reps=20):
    torch.manual_seed(0)
    g = CasNetGenerator((1,) + shape[1:], 6, dimensions=dims, device="cuda")
    x = torch.rand(shape[0], 1, *shape[1:], device="cuda") * 2 - 1
    g.train()
    with torch.no_grad():
        for _ in range(2):
            g(x)                      # give the running statistics something to hold
    g.eval()
    with torch.no_grad():
        for _ in range(3):
            g(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            g(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"eval forward {shape}: {ms:.3f} ms  ({shape[0] / ms * 1e3:.1f} samples/s)")


if __name__ == "__main__":
    run((16, 256, 256), 2)
    run((1, 256, 256), 2)
    run((1, 128, 128, 128), 3)
