#!/usr/bin/env python3
"""Per-call timing of one full G+D step at the bench's C3 shape (development aid).
Every C call of every plan is bracketed by HIP events (engine.KernelProbe(detail=True)), so the
numbers include each call's small helper kernels (reducers, finalizers) and serialise the stream:
use it to rank layers, not to predict the step time."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mpgan_amd import engine
from mpgan_amd.gan import GAN


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--gfwd", action="store_true", help="profile the generator forward only")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    gan = GAN(1, a.size, a.size, dimensions=2, device=dev, g_lr=1e-6, d_lr=1e-6)
    with torch.no_grad():
        gan.discriminator.model_linear[1].weight.mul_(0.02)
    gan.train()
    opts, _ = gan.configure_optimizers()
    batch = bench.synthetic_batch(a.batch, (a.size, a.size), 0, dev)

    def once(i):
        if a.gfwd:
            with torch.no_grad():
                gan.generator(batch["t1w"])
        else:
            gan.fit_batch(batch, i, opts)

    for i in range(2):
        once(i)
    torch.cuda.synchronize()
    probe = engine.KernelProbe(detail=True)
    engine.set_probe(probe)
    for i in range(a.steps):
        once(2 + i)
    torch.cuda.synchronize()
    engine.set_probe(None)
    summ = probe.summary()
    tot = sum(d["ms"] for d in summ.values())
    print(f"sum of timed calls: {tot / a.steps:.2f} ms per {'G forward' if a.gfwd else 'step'}")
    rows = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])
    for k, d in rows[:a.top]:
        per = d["ms"] / a.steps
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] and d["ms"] else 0.0
        print(f"{per:8.3f} ms/step {d['calls'] // a.steps:4d} calls {d['ms'] / d['calls'] * 1e3:8.1f} us/call "
              f"{tf:6.1f} TF  {k}")


if __name__ == "__main__":
    main()
