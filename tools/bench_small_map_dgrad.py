#!/usr/bin/env python3
"""Backward-data of a valid 3^3 conv on a small map (variant B's patch discriminator: 256 -> 512 on 10^3, 896 patches): the
whole launch, each border class alone (MPGAN_DBG_ONLY_PHASE, read per call by the make DEV=1 library) and the
single-phase form (MPGAN_DBG_NO_CLASSES), beside the layer's forward.  Development aid; load the dev library:
  MPGAN_LIB_PATH=$PWD/cross-modality-minipig-gan_amd/libmpgan_hip_dev.so python tools/bench_small_map_dgrad.py"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops


def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=512)
    ap.add_argument("--size", type=int, default=10)
    ap.add_argument("--n", type=int, default=896)
    a = ap.parse_args()
    e, n, cin, cout = a.size, a.n, a.cin, a.cout
    g = ops.ConvGeom(n, (e, e, e), cin, cout, (3, 3, 3), (1, 1, 1), (0, 0, 0))
    o = e - 2
    x = torch.rand(n, e, e, e, cin, device="cuda") * 2 - 1
    y = torch.empty(n, o, o, o, cout, device="cuda")
    dy = torch.rand(n, o, o, o, cout, device="cuda") * 2 - 1
    dx = torch.empty_like(x)
    w = torch.rand(cout, cin, 3, 3, 3, device="cuda") - 0.5
    wp, wpb = ops.pack_weight(w), ops.pack_weight(w, for_dgrad=True)
    real = 2.0 * n * o ** 3 * cin * cout * 27
    ms = t(lambda: ops.conv_forward(g, x, wp, None, y))
    print(f"forward                      {ms:8.2f} ms  {real / ms / 1e9:7.1f} TFLOP/s")
    ms = t(lambda: ops.conv_backward_data(g, dy, wpb, dx))
    print(f"backward-data, all classes   {ms:8.2f} ms  {real / ms / 1e9:7.1f} TFLOP/s algorithmic")
    # classes per dimension: left border (2 rows, taps 0..1 of which one is masked for the outermost row), interior, right border
    ext = [2, e - 4, 2]
    taps = [2, 3, 2]
    tot = 0.0
    i = 0
    for cz in range(3):
        for cy in range(3):
            for cx in range(3):
                os.environ["MPGAN_DBG_ONLY_PHASE"] = str(i)
                ms = t(lambda: ops.conv_backward_data(g, dy, wpb, dx))
                m = n * ext[cz] * ext[cy] * ext[cx]
                issued = 2.0 * m * taps[cz] * taps[cy] * taps[cx] * cin * cout
                tot += ms
                print(f"  class {i:2d} ({cz}{cy}{cx}): {m:8d} rows x {taps[cz] * taps[cy] * taps[cx]:2d} taps  {ms:7.2f} ms  "
                      f"{issued / ms / 1e9:7.1f} TFLOP/s issued")
                i += 1
    del os.environ["MPGAN_DBG_ONLY_PHASE"]
    print(f"sum of the classes           {tot:8.2f} ms")
    os.environ["MPGAN_DBG_NO_CLASSES"] = "1"
    ms = t(lambda: ops.conv_backward_data(g, dy, wpb, dx))
    print(f"backward-data, one phase     {ms:8.2f} ms  {real / ms / 1e9:7.1f} TFLOP/s algorithmic, "
          f"{2.0 * n * e ** 3 * cin * cout * 27 / ms / 1e9:7.1f} issued")


if __name__ == "__main__":
    main()
