#!/usr/bin/env python3
"""Experiment: replaying a generator plan's programs as a captured HIP graph vs. eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.networks import CasNetGenerator

dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = CasNetGenerator((1, 256, 256), 6, dimensions=2, device=dev)
g.train()
x = torch.rand(16, 1, 256, 256, device=dev) * 2 - 1
x.requires_grad_(True)
for _ in range(3):
    y = g(x)
    y.sum().backward()
torch.cuda.synchronize()
plan = [pl for pool in g._plans.values() for pl in pool if pl.want_backward][0]


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


print("eager fwd %.3f ms   bwd %.3f ms" % (timeit(plan.fwd.run), timeit(plan.bwd.run)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
for name, prog in (("fwd", plan.fwd), ("bwd", plan.bwd)):
    try:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            prog.run()
        torch.cuda.synchronize()
        print("graph %s %.3f ms" % (name, timeit(gr.replay)))
    except Exception as e:
        print("graph", name, "failed:", repr(e)[:300])
