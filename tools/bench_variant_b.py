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
    ap.add_argument("--families", action="store_true",
                    help="per-call HIP-event timing of the conv launches (single stream): TFLOP/s per kernel family and "
                         "per layer -- the 64..512-channel 16^3 convs of the patch discriminator, its 262,144-input head")
    a = ap.parse_args()
    if a.families:
        os.environ["MPGAN_SINGLE_STREAM"] = "1"
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
    if a.families:
        from mpgan_amd import engine
        probe = engine.KernelProbe(detail=True)
        engine.set_probe(probe)
        for i in range(a.steps):
            gan.fit_batch(batch, 100 + i, opts)
        torch.cuda.synchronize()
        engine.set_probe(None)
        summ = probe.summary()
        fam = {}
        for k, d in summ.items():
            if not d["flops"]:
                continue
            f = "wgrad" if k.startswith("conv_backward_weight") else ("dgrad" if k.startswith("conv_backward_data") else "fwd")
            e = fam.setdefault(f, dict(ms=0.0, flops=0.0, calls=0))
            e["ms"] += d["ms"]; e["flops"] += d["flops"]; e["calls"] += d["calls"]
        print("conv launches per step (HIP events on the launch stream, single stream; fp32 matrix peak 157.3 TFLOP/s):")
        for f, e in sorted(fam.items()):
            tf = e["flops"] / (e["ms"] * 1e-3) / 1e12
            print(f"  {f:6s} {e['calls'] // a.steps:5d} launches {e['ms'] / a.steps:9.2f} ms/step {tf:7.1f} TFLOP/s = {tf / 157.3:.3f} of peak")
        print("  --- layers by time")
        for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:24]:
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
            print(f"  {d['ms'] / a.steps:9.3f} ms/step {d['calls'] // a.steps:4d} x {d['ms'] / d['calls'] * 1e3:9.1f} us {tf:7.1f} TFLOP/s  {k}")
    print(f"variant B: {a.batch} x {a.size}^3 volumes, {a.batch * a.samples} patches per step: {dt * 1e3:.1f} ms/step, "
          f"{a.batch / dt:.2f} volumes/s; losses " + ", ".join(f"{k}={float(v):.4f}" for k, v in logs.items()))


if __name__ == "__main__":
    main()
