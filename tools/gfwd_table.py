#!/usr/bin/env python3
"""DESIGN.md's table of the generator's conv layers against their roofline floor:
floor = max(algorithmic FLOP / 157.3 TFLOP/s, algorithmic bytes / 8 TB/s); bytes = input + output read / written once
+ weights.  Reads a tools/bench_conv.py table (profiles/rNN_layer_bench.txt)."""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

LAYERS = {   # as tools/bench_conv.py: (cin, cout, k, stride, pad, in_hw, transposed); launches per U-Net forward
    "G.d0.u0": (1, 16, 3, 2, 1, 256, False), "G.d0.u1": (16, 16, 3, 1, 1, 128, False),
    "G.d1.u0": (16, 32, 3, 2, 1, 128, False), "G.d1.u1": (32, 32, 3, 1, 1, 64, False),
    "G.d2.u0": (32, 64, 3, 2, 1, 64, False), "G.d2.u1": (64, 64, 3, 1, 1, 32, False),
    "G.b.u0": (64, 128, 3, 1, 1, 32, False), "G.b.u1": (128, 128, 3, 1, 1, 32, False),
    "G.b.res": (64, 128, 1, 1, 0, 32, False), "G.up2.T": (192, 32, 3, 2, 1, 32, True),
    "G.up1.T": (64, 16, 3, 2, 1, 64, True), "G.up0.T": (32, 1, 3, 2, 1, 128, True), "G.up0.ru": (1, 1, 3, 1, 1, 256, False),
}


def main(path, n=16):
    rows = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+([\d.]+) GF\s+fwd\s+([\d.]+) us.*dgrad\s+([\d.]+) us.*wgrad\s+([\d.]+) us", line)
        if m:
            rows[m.group(1)] = tuple(float(m.group(i)) for i in (2, 3, 4, 5))
    print("| layer | shape | GFLOP | MB | floor µs (MFMA / HBM) | fwd µs | frac | dgrad µs | wgrad µs |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, (cin, cout, k, s, p, hw, tr) in LAYERS.items():
        if name not in rows:
            continue
        gf, fwd, dg, wg = rows[name]
        ohw = hw * s if tr else (hw + 2 * p - k) // s + 1
        mb = (n * hw * hw * cin + n * ohw * ohw * cout + cin * cout * k * k) * 4 / 1e6
        f_m, f_h = gf * 1e9 / 157.3e12 * 1e6, mb * 1e6 / 8e12 * 1e6
        fl = max(f_m, f_h)
        print(f"| {name} | {cin}→{cout} k{k} s{s}{'T' if tr else ''} @{hw}² | {gf:.2f} | {mb:.1f} | {f_m:.1f} / {f_h:.1f} | {fwd:.1f} | "
              f"{fl / fwd:.2f} | {dg:.1f} | {wg:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
