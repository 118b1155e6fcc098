#!/usr/bin/env python3
"""The fused generator forward alone (BASELINE config C2: 256x256, bs 16, fp32, train-mode BatchNorm), N times
back to back -- the program to put behind `rocprofv3 --kernel-trace --stats --` for profiles/rNN_gfwd_kernel_stats.csv.
Prints the wall time per forward."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mpgan_amd.gan import GAN


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--norm", default="batch")
    ap.add_argument("--backward", action="store_true",
                    help="forward + backward (L1-type upstream gradient) per repetition: the generator's kernels of a "
                         "training step without the discriminator (profiles/rNN_gbwd_kernel_stats.csv)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    gan = GAN(1, a.size, a.size, dimensions=2, device=dev, norm=a.norm)
    gan.train()
    x = bench.synthetic_batch(a.batch, (a.size, a.size), 0, dev)["t1w"]
    if a.backward:
        from mpgan_amd import engine
        engine._SINGLE_STREAM = True                              # weight gradients on the caller's stream: durations add up
        for i in range(3 + a.reps):
            if i == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            gan.generator.zero_grad()
            gan.generator(x).abs().mean().backward()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        print(f"G forward + backward {a.size}x{a.size} bs{a.batch}: {ms:.3f} ms")
        return
    with torch.no_grad():
        for _ in range(3):
            gan.generator(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            gan.generator(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    print(f"G forward {a.size}x{a.size} bs{a.batch}: {ms:.3f} ms = {a.batch / ms * 1e3:.0f} slices/s "
          f"({7.2423e9 * (a.size / 256) ** 2 * a.batch / (ms * 1e-3) / 157.3e12:.3f} of the fp32 matrix peak)")


if __name__ == "__main__":
    main()
