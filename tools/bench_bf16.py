#!/usr/bin/env python3
"""GPU time per launch of the bf16-storage discriminator's dense layers (config C5: 128^3, bs 4; or 2-D 256^2 bs 16):
forward (with statistics rows), backward-data, weight gradient.  Development aid."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops

LAYERS3 = {"D.conv2": (64, 128, 3, 1, 126), "D.conv3": (128, 256, 4, 2, 124), "D.conv4": (256, 256, 4, 2, 61)}
LAYERS2 = {"D.conv2": (64, 128, 3, 1, 254), "D.conv3": (128, 256, 4, 2, 252), "D.conv4": (256, 256, 4, 2, 125)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, default=3)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--layers", default="D.conv2,D.conv3,D.conv4")
    ap.add_argument("--modes", default="fwd,dgrad,wgrad")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    n = a.n or (4 if a.dims == 3 else 16)
    table = LAYERS3 if a.dims == 3 else LAYERS2
    BF = torch.bfloat16
    for name in a.layers.split(","):
        cin, cout, k, s, e = table[name]
        sp = (e, e, e) if a.dims == 3 else (1, e, e)
        kk = (k, k, k) if a.dims == 3 else (1, k, k)
        ss = (s, s, s) if a.dims == 3 else (1, s, s)
        g = ops.ConvGeom(n, sp, cin, cout, kk, ss, (0, 0, 0), False, (0, 0, 0))
        x = (torch.rand(n, *sp, cin, device="cuda") * 2 - 1).to(BF)
        y = torch.empty(n, *g.out_dhw, cout, device="cuda", dtype=BF)
        dy = (torch.rand(n, *g.out_dhw, cout, device="cuda") * 2 - 1).to(BF)
        dx = torch.empty_like(x)
        w = torch.rand(cout, cin, *kk[3 - a.dims:], device="cuda") - 0.5
        wp, wpb = ops.pack_weight_bf16(w), ops.pack_weight_bf16(w, for_dgrad=True)
        bias = torch.rand(cout, device="cuda")
        part = torch.empty(ops.conv_stats_rows_bf16(g) * 2 * cout, device="cuda")
        dw = torch.empty_like(w)
        ws = torch.empty(ops.conv_wgrad_workspace_bf16(g) // 4 + 1, device="cuda")
        taps = k ** a.dims
        flops = 2.0 * n * g.out_dhw[0] * g.out_dhw[1] * g.out_dhw[2] * cin * cout * taps
        out = [f"{name:8s} {flops / 1e9:8.1f} GF"]
        for mode in a.modes.split(","):
            fn = {"fwd": lambda: ops.conv_forward_bf16(g, x, wp, bias, y, stats_partials=part),
                  "dgrad": lambda: ops.conv_backward_data_bf16(g, dy, wpb, dx),
                  "wgrad": lambda: ops.conv_backward_weight_bf16(g, x, dy, dw, ws)}[mode]
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            out.append(f"{mode} {us:8.1f} us {flops / us / 1e6:6.1f} TF")
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
