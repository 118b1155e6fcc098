#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the BASELINE C3 shapes (256x256, bs 16).
Development aid: prints TFLOP/s (algorithmic) per layer for fwd / dgrad / wgrad."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops

LAYERS = {
    # name: (cin, cout, k, stride, pad, in_hw, transposed)
    "D.conv1": (1, 64, 3, 1, 0, 256, False),
    "D.conv2": (64, 128, 3, 1, 0, 254, False),
    "D.conv3": (128, 256, 4, 2, 0, 252, False),
    "D.conv4": (256, 256, 4, 2, 0, 125, False),
    "G.d0.u0": (1, 16, 3, 2, 1, 256, False),
    "G.d0.u1": (16, 16, 3, 1, 1, 128, False),
    "G.d1.u0": (16, 32, 3, 2, 1, 128, False),
    "G.d1.u1": (32, 32, 3, 1, 1, 64, False),
    "G.d2.u0": (32, 64, 3, 2, 1, 64, False),
    "G.d2.u1": (64, 64, 3, 1, 1, 32, False),
    "G.b.u0": (64, 128, 3, 1, 1, 32, False),
    "G.b.u1": (128, 128, 3, 1, 1, 32, False),
    "G.b.res": (64, 128, 1, 1, 0, 32, False),
    "G.up2.T": (192, 32, 3, 2, 1, 32, True),
    "G.up1.T": (64, 16, 3, 2, 1, 64, True),
    "G.up0.T": (32, 1, 3, 2, 1, 128, True),
    "G.up0.ru": (1, 1, 3, 1, 1, 256, False),
}


def time_it(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="D.conv2,D.conv3,D.conv4")
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--modes", default="fwd,dgrad,wgrad")
    ap.add_argument("--pro", action="store_true", help="apply a BN+LeakyReLU prologue like the network does")
    a = ap.parse_args()
    names = list(LAYERS) if a.layers == "all" else a.layers.split(",")
    for name in names:
        cin, cout, k, s, p, hw, tr = LAYERS[name]
        g = ops.ConvGeom(a.n, (1, hw, hw), cin, cout, (1, k, k), (1, s, s), (0, p, p), tr, (0, s - 1, s - 1) if tr else (0, 0, 0))
        x = torch.rand(a.n, 1, hw, hw, cin, device="cuda") * 2 - 1
        y = torch.empty(a.n, *g.out_dhw, cout, device="cuda")
        dy = torch.rand(a.n, *g.out_dhw, cout, device="cuda") * 2 - 1
        dx = torch.empty_like(x)
        w = (torch.rand(cin, cout, k, k, device="cuda") if tr else torch.rand(cout, cin, k, k, device="cuda")) - 0.5
        wp = ops.pack_weight(w, transposed=tr)
        wpb = ops.pack_weight(w, transposed=tr, for_dgrad=True)
        dw = torch.empty_like(w)
        ws = torch.empty(max(ops.conv_wgrad_workspace(g) // 4, 4), device="cuda")
        bias = torch.rand(cout, device="cuda")
        pro = None
        if a.pro and not tr:
            pro = ops.Prologue(torch.rand(cin, device="cuda") + 0.5, torch.rand(cin, device="cuda") - 0.5, 0, ops.ACT_LEAKY, 0.2)
        grid = g.in_dhw if tr else g.out_dhw
        flops = 2.0 * a.n * grid[1] * grid[2] * cin * cout * k * k
        out = [f"{name:9s} {flops/1e9:8.2f} GF"]
        for mode in a.modes.split(","):
            if mode == "fwd":
                ms = time_it(lambda: ops.conv_forward(g, x, wp, bias, y, pro=pro), a.reps)
            elif mode == "dgrad":
                ms = time_it(lambda: ops.conv_backward_data(g, dy, wpb, dx), a.reps)
            else:
                ms = time_it(lambda: ops.conv_backward_weight(g, x, dy, dw, ws, pro=pro), a.reps)
            out.append(f"{mode} {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF")
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
