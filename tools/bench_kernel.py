#!/usr/bin/env python3
"""GPU time per launch of one conv layer's kernels, measured on a pre-built engine Program that repeats the launch
(ctypes calls on frozen arguments, ~3 us of host time each) -- tools/bench_conv.py goes through the Python op
wrappers (~15 us of host time per call) and is host-bound below that.  Development aid for the what-if builds
(MPGAN_DBG_PATCH_SKIP and friends)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import engine, ops
from bench_conv import LAYERS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="G.d0.u1")
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--modes", default="fwd,dgrad")
    ap.add_argument("--pro", action="store_true")
    ap.add_argument("--stats", action="store_true", help="forward with fused statistics partial rows")
    a = ap.parse_args()
    for name in a.layers.split(","):
        cin, cout, k, s, p, hw, tr = LAYERS[name]
        g = ops.ConvGeom(a.n, (1, hw, hw), cin, cout, (1, k, k), (1, s, s), (0, p, p), tr, (0, s - 1, s - 1) if tr else (0, 0, 0))
        x = torch.rand(a.n, 1, hw, hw, cin, device="cuda") * 2 - 1
        y = torch.empty(a.n, *g.out_dhw, cout, device="cuda")
        dy = torch.rand(a.n, *g.out_dhw, cout, device="cuda") * 2 - 1
        dx = torch.empty_like(x)
        w = (torch.rand(cin, cout, k, k, device="cuda") if tr else torch.rand(cout, cin, k, k, device="cuda")) - 0.5
        wp = ops.pack_weight(w, transposed=tr)
        wpb = ops.pack_weight(w, transposed=tr, for_dgrad=True)
        bias = torch.rand(cout, device="cuda")
        pro = None
        if a.pro and not tr:
            pro = ops.Prologue(torch.rand(cin, device="cuda") + 0.5, torch.rand(cin, device="cuda") - 0.5, 0, ops.ACT_LEAKY,
                               1.0, torch.tensor([0.25], device="cuda"))
        part = None
        if a.stats:
            rows = ops.conv_stats_rows(g, 1 if pro is not None else 0)
            part = torch.empty((rows + 32) * 2 * cout, device="cuda") if rows else None
        grid = g.in_dhw if tr else g.out_dhw
        flops = 2.0 * a.n * grid[1] * grid[2] * cin * cout * k * k
        out = [f"{name:9s} {flops / 1e9:7.2f} GF"]
        for mode in a.modes.split(","):
            prog = engine.Program()
            for _ in range(a.reps):
                if mode == "fwd":
                    engine.emit_conv_fwd(prog, g, x, wp, bias, y, pro=pro, stats=part)
                else:
                    engine.emit_conv_dgrad(prog, g, dy, wpb, dx)
            prog.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prog.run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            out.append(f"{mode} {us:7.1f} us {flops / us / 1e6:6.1f} TF")
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
