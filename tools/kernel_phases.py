#!/usr/bin/env python3
"""MEASURED in-kernel time breakdown of the generator's conv launches (development aid).

Runs the generator's forward (default) or backward at the bench's shape on the stamps build of the library
(`make -C cross-modality-minipig-gan_amd/csrc STAMPS=1` -> libmpgan_hip_stamps.so): thread 0 of every block
records the 100 MHz device clock at its phase boundaries (csrc/mpgan_common.h: MPGAN_STAMP).  Per launch this
prints, in microseconds,
    span     first block's start -> last block's end (what rocprofv3 reports as the kernel, minus dispatch)
    ramp     95th-percentile block start after the first one (dispatch ramp, or the second round of blocks)
    K-stepped kernel:   prologue (first tile's loads -> LDS) | K loop | in-block split-K fold | epilogue = row->pixel
                        map + output stores issued + statistics rows
    persistent patch:   weight staging | first patch | sum over the block's tiles of contraction + epilogue (of
                        which: residual prefetch + fragment reads + MFMAs) | sum of tile tails (barrier, statistics
                        rows, next patch's arrival + LDS stores) | tiles per block
    gap      this launch's last end -> the next stamped launch's first start (seam + any unstamped kernels
             between: BatchNorm finalize, the residual-sum pass)
as medians over the blocks, averaged over the six U-Nets of the cascade.

    python tools/kernel_phases.py [--what fwd|bwd] [--size 256] [--batch 16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "cross-modality-minipig-gan_amd", "libmpgan_hip_stamps.so")
os.environ.setdefault("MPGAN_LIB_PATH", STAMPS_LIB)
os.environ.setdefault("MPGAN_SINGLE_STREAM", "1")

import numpy as np      # noqa: E402
import torch            # noqa: E402

SLOTS, CAP = 12, 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="fwd", choices=("fwd", "bwd"))
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--all", action="store_true", help="print every launch instead of the average over the U-Nets")
    a = ap.parse_args()
    if not os.path.exists(os.environ["MPGAN_LIB_PATH"]):
        raise SystemExit("build the stamps library first: make -C cross-modality-minipig-gan_amd/csrc STAMPS=1")
    from mpgan_amd._lib import lib
    from mpgan_amd.networks import CasNetGenerator
    L = lib()
    khz = L.mpgan_debug_clock_khz()
    tick_us = 1e3 / khz
    sp = (a.size,) * a.dims
    torch.manual_seed(0)
    g = CasNetGenerator((1, *sp), 6, dimensions=a.dims, device="cuda").train()
    x = (torch.rand(a.batch, 1, *sp, device="cuda") * 2 - 1).requires_grad_(a.what == "bwd")
    for _ in range(3):
        if a.what == "bwd":
            g(x).abs().mean().backward()
        else:
            with torch.no_grad():
                g(x)
    torch.cuda.synchronize()
    plan = next(pl for pool in g._plans.values() for pl in pool if pl.want_backward == (a.what == "bwd"))
    prog = plan.fwd if a.what == "fwd" else plan.bwd
    stamped = ("conv_forward", "conv_backward_data")
    calls = [(prog.names[i], prog.descs[i], (prog.tags[i] or ("?",))[0]) for i in range(len(prog)) if prog.names[i] in stamped]
    nl = len(calls)
    buf = torch.zeros(nl * CAP * SLOTS, dtype=torch.int64, device="cuda")
    if a.what == "bwd":
        loss = g(x).abs().mean()                        # (the forward in front of it is not stamped)
        torch.cuda.synchronize()
        assert L.mpgan_debug_stamps(buf.data_ptr(), nl, CAP) == 0, L.mpgan_last_error().decode()
        loss.backward()
    else:
        torch.cuda.synchronize()
        buf.zero_()
        assert L.mpgan_debug_stamps(buf.data_ptr(), nl, CAP) == 0, L.mpgan_last_error().decode()
        with torch.no_grad():
            g(x)
    torch.cuda.synchronize()
    used = L.mpgan_debug_stamps_used()
    L.mpgan_debug_stamps(None, 0, 0)
    assert used == nl, (used, nl)
    s = buf.cpu().numpy().reshape(nl, CAP, SLOTS).astype(np.float64)
    rows = []
    for i, (name, desc, kern) in enumerate(calls):
        blk = s[i][s[i][:, 0] > 0]
        if len(blk) == 0:
            rows.append(dict(key=(name, desc, kern), kind=0, span=np.nan, start=np.nan, end=np.nan))
            continue
        kind = int(blk[0, 6])
        t0, t7 = blk[:, 0], blk[:, 7]
        done = t7 > 0
        r = dict(key=(name, desc, kern), kind=kind, blocks=len(blk), start=t0.min(), end=t7[done].max() if done.any() else np.nan)
        r["span"] = (r["end"] - r["start"]) * tick_us
        r["ramp"] = (np.percentile(t0, 95) - t0.min()) * tick_us
        med = lambda v: float(np.median(v)) * tick_us
        b = blk[done]
        if kind == 1:
            r.update(prologue=med(b[:, 1] - b[:, 0]), kloop=med(b[:, 2] - b[:, 1]), fold=med(b[:, 3] - b[:, 2]),
                     epilogue=med(b[:, 7] - b[:, 3]), ep_map=med(b[:, 4] - b[:, 3]), ep_store=med(b[:, 5] - b[:, 4]),
                     ep_stats=med(b[:, 7] - b[:, 5]), block=med(b[:, 7] - b[:, 0]))
        elif kind == 2:
            r.update(wstage=med(b[:, 1] - b[:, 0]), patch0=med(b[:, 2] - b[:, 1]), contract=med(b[:, 3]), tails=med(b[:, 4]),
                     mfma=med(b[:, 8]), tiles=float(np.median(b[:, 5])), block=med(b[:, 7] - b[:, 0]))
        rows.append(r)
    for i, r in enumerate(rows):
        nxt = rows[i + 1]["start"] if i + 1 < nl else np.nan
        r["gap"] = (nxt - r["end"]) * tick_us
    per = nl // 6
    print(f"{a.what}: {nl} stamped launches ({per} per U-Net), clock {khz} kHz; microseconds, medians over blocks"
          + ("" if a.all else ", mean over the six U-Nets"))
    hdr = f"{'launch':58s} {'blocks':>6s} {'span':>6s} {'ramp':>5s} | {'phases':92s} | {'gap':>5s}"
    print(hdr)
    groups = [[rows[i]] for i in range(nl)] if a.all else [[rows[u * per + j] for u in range(6)] for j in range(per)]
    tot_span = tot_gap = 0.0
    for grp in groups:
        r0 = grp[0]
        mean = lambda k: float(np.nanmean([r.get(k, np.nan) for r in grp]))
        name, desc, kern = r0["key"]
        label = f"{'fwd' if name == 'conv_forward' else 'dgrad'} {desc} {kern.replace('gather_', '').replace('_kernel', '')}"[:58]
        if r0["kind"] == 1:
            ph = (f"prologue {mean('prologue'):4.1f}  K loop {mean('kloop'):5.1f}  fold {mean('fold'):3.1f}  "
                  f"epilogue {mean('epilogue'):4.1f} = map {mean('ep_map'):3.1f} + stores {mean('ep_store'):3.1f} + stats {mean('ep_stats'):3.1f}  "
                  f"(block {mean('block'):5.1f})")
        elif r0["kind"] == 2:
            ph = (f"weights {mean('wstage'):4.1f}  patch0 {mean('patch0'):4.1f}  contract {mean('contract'):5.1f} "
                  f"(MFMA loops {mean('mfma'):4.1f})  tails {mean('tails'):4.1f}  tiles {mean('tiles'):3.0f}  (block {mean('block'):5.1f})")
        else:
            ph = "(no stamps in this kernel)"
        span, gap = mean("span"), mean("gap")
        if span == span:
            tot_span += span
        if gap == gap:
            tot_gap += gap
        print(f"{label:58s} {mean('blocks') if r0['kind'] else float('nan'):6.0f} {span:6.1f} {mean('ramp') if r0['kind'] else float('nan'):5.1f} | {ph:92s} | {gap:5.1f}")
    scale = 1 if a.all else 6
    print(f"sum of spans {tot_span * scale / 1e3:.3f} ms, sum of gaps {tot_gap * scale / 1e3:.3f} ms per pass")


if __name__ == "__main__":
    main()
