#!/usr/bin/env python3
"""Wall time of the phases of one G+D step at the bench's shape, single stream, HIP events between phases
(development aid: where the step's time goes at the granularity of network passes)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPGAN_SINGLE_STREAM", "1")
import torch
import bench
from mpgan_amd import engine
from mpgan_amd.gan import GAN, adversarial_loss, reconstruction_loss, scalar_axpby


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--probe", default="", help="gfwd | gbwd | dbwd: per-call HIP-event timing of that phase's calls")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    sp = (a.size,) * a.dims
    gan = GAN(1, *sp, dimensions=a.dims, device=dev, g_lr=1e-6, d_lr=1e-6, storage_dtype=a.dtype)
    with torch.no_grad():
        gan.discriminator.model_linear[1].weight.mul_(0.02)
    gan.train()
    opt_g, opt_d = gan.configure_optimizers()[0]
    batch = bench.synthetic_batch(a.batch, sp, 0, dev)
    x, t = batch["t1w"], batch["t2w"]
    G, D = gan.generator, gan.discriminator
    marks = []
    probe = engine.KernelProbe(detail=True)
    probing = [False]

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    def step():
        for p in D.parameters():
            p.requires_grad_(False)
        for p in G.parameters():
            p.requires_grad_(True)
        opt_g.zero_grad()
        mark("start")
        if a.probe == "gfwd" and probing[0]:
            engine.set_probe(probe)
        y = G(x)
        if a.probe == "gfwd":
            engine.set_probe(None)
        mark("G fwd (train, grads)")
        def at_y(g):
            mark("D bwd (input grad only) + L1 bwd")
            if a.probe == "gbwd" and probing[0]:
                engine.set_probe(probe)
        y.register_hook(at_y)
        p = D(y)
        mark("D fwd (fake)")
        valid = torch.ones(a.batch, 1, device=dev)
        loss = scalar_axpby(adversarial_loss(p, valid), 1.0, reconstruction_loss(y, t), 1.0)
        mark("losses")
        loss.backward()
        if a.probe == "gbwd":
            engine.set_probe(None)
        mark("G bwd")
        opt_g.step()
        mark("Adam G")
        for p_ in D.parameters():
            p_.requires_grad_(True)
        for p_ in G.parameters():
            p_.requires_grad_(False)
        opt_d.zero_grad()
        mark("zero_grad D")
        with torch.no_grad():
            y2 = G(x)
        mark("G fwd (no grad)")
        lr = adversarial_loss(D(t), torch.full((a.batch, 1), 0.9, device=dev))
        mark("D fwd (real)")
        lf = adversarial_loss(D(y2), torch.zeros(a.batch, 1, device=dev))
        mark("D fwd (fake)")
        d_loss = scalar_axpby(lr, 0.5, lf, 0.5)
        if a.probe == "dbwd" and probing[0]:
            engine.set_probe(probe)
        d_loss.backward()
        engine.set_probe(None)
        mark("D bwd x2 (dgrad + wgrad)")
        opt_d.step()
        mark("Adam D")

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    probing[0] = True
    tot = {}
    order = []
    for _ in range(a.steps):
        marks.clear()
        step()
        torch.cuda.synchronize()
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            key = f"{len(order):02d} {n1}" if len(order) < len(marks) - 1 else None
            if key is not None:
                order.append(key)
        for i, ((n0, e0), (n1, e1)) in enumerate(zip(marks[:-1], marks[1:])):
            tot[order[i]] = tot.get(order[i], 0.0) + e0.elapsed_time(e1)
    s = 0.0
    for k in order:
        ms = tot[k] / a.steps
        s += ms
        print(f"{ms:8.3f} ms  {k}")
    print(f"{s:8.3f} ms  total (single stream)")
    if a.probe:
        summ = probe.summary()
        tot_p = sum(d["ms"] for d in summ.values())
        print(f"probed calls of {a.probe}: {tot_p / a.steps:.2f} ms per step")
        byname = {}
        for k, d in summ.items():
            nm = k.split("|")[0].split(" ")[0]
            e = byname.setdefault(nm, [0.0, 0])
            e[0] += d["ms"]; e[1] += d["calls"]
        for nm, (ms, calls) in sorted(byname.items(), key=lambda kv: -kv[1][0]):
            print(f"{ms / a.steps:8.3f} ms/step {calls // a.steps:4d} calls {ms / calls * 1e3:8.1f} us/call  {nm}")
        print("--- top calls")
        for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:40]:
            print(f"{d['ms'] / a.steps:8.3f} ms/step {d['calls'] // a.steps:4d} calls {d['ms'] / d['calls'] * 1e3:8.1f} us/call  {k}")


if __name__ == "__main__":
    main()
