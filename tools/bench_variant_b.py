#!/usr/bin/env python3
"""Throughput of the reference's variant B (test_runs/GAN.py: 4-U-Net generator with channels
(32, 64, 128, 256) on 128^3 volumes, patch discriminator on 128 random 16^3 crops per volume with the
perceptual loss over its 16 taps) -- SURVEY.md section 8(f) row N4.  Development aid, not the headline."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.gan_patch import GAN


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2, help="volumes per step (the reference used 7)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--samples", type=int, default=128, help="16^3 crops per volume")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    gan = GAN(1, a.size, a.size, a.size, num_samples=a.samples, crop_seed=1, lr=1e-6)
    gan.train()
    opts, _ = gan.configure_optimizers()
    gen = torch.Generator().manual_seed(1)
    batch = {k: (torch.rand(a.batch, 1, a.size, a.size, a.size, generator=gen) * 2 - 1).to(dev) for k in ("t1w", "t2w")}
    gan.fit_batch(batch, 0, opts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        logs = gan.fit_batch(batch, 1 + i, opts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"variant B: {a.batch} x {a.size}^3 volumes, {a.batch * a.samples} patches per step: {dt * 1e3:.1f} ms/step, "
          f"{a.batch / dt:.2f} volumes/s; losses " + ", ".join(f"{k}={float(v):.4f}" for k, v in logs.items()))


if __name__ == "__main__":
    main()
