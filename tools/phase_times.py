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
from mpgan_amd.gan import GAN


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
    probe = engine.KernelProbe(detail=True)

    def on_phase(name, starting):
        if name == a.probe:
            engine.set_probe(probe if starting else None)

    for i in range(2):
        gan.fit_batch(batch, i, [opt_g, opt_d])
    torch.cuda.synchronize()
    rows = bench.phase_breakdown(gan, [opt_g, opt_d], batch, steps=a.steps, on_phase=on_phase if a.probe else None)
    engine.set_probe(None)
    order = [f"{i:02d} {nm}" for i, (nm, _, _, _) in enumerate(rows)]
    tot = {k: r[1] * a.steps for k, r in zip(order, rows)}
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
        print("--- top calls (conv launches: algorithmic GB/s = gathered + dense operand once over the launch time, and TFLOP/s)")
        hbm_bound_ms = 0.0
        for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:60]:
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["bytes"] else 0.0
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
            extra = f"  {gbs:7.0f} GB/s {tf:6.1f} TF/s" if d["bytes"] else ""
            print(f"{d['ms'] / a.steps:8.3f} ms/step {d['calls'] // a.steps:4d} calls {d['ms'] / d['calls'] * 1e3:8.1f} us/call  {k}{extra}")
        # what bf16 storage of these tensors could save at most: a launch's time can only follow its bytes where it
        # already runs near the HBM roof; take time * min(1, GB/s / 4000) as its bandwidth-bound share and halve that
        for k, d in summ.items():
            if d["bytes"]:
                gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                hbm_bound_ms += d["ms"] / a.steps * min(1.0, gbs / 4000.0) * 0.5
        print(f"upper bound of what halving the conv operands' bytes (bf16 storage) could save in this phase: "
              f"{hbm_bound_ms:.2f} ms per step (launch time x min(1, algorithmic GB/s / 4000) / 2, summed)")


if __name__ == "__main__":
    main()
