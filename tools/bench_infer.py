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
    # the same program replayed as a captured HIP graph (no Python / ctypes between the launches) and the
    # train-mode forward beside it: is the eager figure the GPU's or the host's?
    plan = [pl for key, pool in g._plans.items() for pl in pool if key[4] is False][0]      # key[4]: gen.training
    n_launch = sum(1 for fn, _ in plan.fwd.calls if fn is not None)
    gms = float("nan")
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            plan.fwd.run()
        torch.cuda.synchronize()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t) / reps * 1e3
    except Exception as e:                      # noqa: BLE001 (development aid)
        print("graph capture failed:", repr(e)[:200])
    g.train()
    with torch.no_grad():
        for _ in range(3):
            g(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            g(x)
        torch.cuda.synchronize()
    tms = (time.perf_counter() - t) / reps * 1e3
    print(f"eval forward {shape}: {ms:.3f} ms eager ({n_launch} launches), {gms:.3f} ms as a HIP graph; train-mode forward "
          f"{tms:.3f} ms  ({shape[0] / ms * 1e3:.1f} samples/s eager)")


if __name__ == "__main__":
    run((16, 256, 256), 2)
    run((1, 256, 256), 2)
    run((1, 128, 128, 128), 3)
