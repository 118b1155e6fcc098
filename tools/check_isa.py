#!/usr/bin/env python3
"""Build-time check of the LDS-DMA kernels' K loops (bf16: conv_bf16; fp32: gather_conv_dma_kernel in conv_igemm) (run by __graft_entry__.build()): the LDS-DMA pipeline only
works if the loop body waits with the COUNTED vmcnt it was written with.  hipcc inserts `s_waitcnt vmcnt(0)`
in front of any LDS access it thinks may alias an LDS-DMA in flight, silently serialising the pipeline; this
script reads the device assembly hipcc leaves beside the object (-save-temps=obj) and fails if the innermost
loop of a pipelined kernel holds a vmcnt(0), scratch traffic, or fewer MFMAs / DMA loads than expected."""
import re
import sys

MFMA_BF16, MFMA_F32 = "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32"
KERNELS_F32 = {   # conv_igemm: the DMA-staged fp32 form <BN, TM, TN, WN, NST> (64 fp32 MFMAs per 128 x 128 K-step and wave)
    "gather_conv_dma_kernelILi128ELi2ELi2ELi2ELi2": (64, 4),
    "gather_conv_dma_kernelILi64ELi1ELi2ELi1ELi2": (32, 3),
    "gather_conv_dma_kernelILi32ELi1ELi1ELi1ELi2": (16, 2),
}
KERNELS = {   # mangled-name fragment -> (min MFMAs, min LDS-DMA loads) in the K loop
    "gather_conv_bf16_kernelILi128ELb0ELi4": (32, 12),
    "gather_conv_bf16_kernelILi128ELb1ELi4": (32, 12),
    "gather_conv_bf16_kernelILi64ELb0ELi4": (16, 10),
    "gather_conv_bf16_kernelILi64ELb1ELi4": (16, 10),
    "gather_conv_bf16_kernelILi128ELb0ELi8": (16, 6),     # eight waves: half the tile and half the loads per wave
    "gather_conv_bf16_kernelILi128ELb1ELi8": (16, 6),
    "gather_conv_bf16_kernelILi64ELb0ELi8": (8, 5),
    "gather_conv_bf16_kernelILi64ELb1ELi8": (8, 5),
    "gather_patch_bf16_kernelILi128": (16, 2),            # patch form: weights only stream inside the tap loop
    "gather_patch_bf16_kernelILi64": (8, 1),
    "gather_patch8_bf16_kernelILi128": (16, 3),           # big-patch form: a tap = two k-subs of 8 (4) MFMAs, one group of weights
    "gather_patch8_bf16_kernelILi64": (8, 2),
    "gather_conv_bf16_wide_kernelILi2ELi4ELb0ELb1ELb0": (32, 8),   # wide form <WM, WN, MASK, RING, PAIR>: 128 x 64 per wave
    "gather_conv_bf16_wide_kernelILi2ELi4ELb1ELb1ELb0": (32, 8),
    "gather_conv_bf16_wide_kernelILi2ELi4ELb0ELb1ELb1": (32, 8),   # ... over phase pairs
    "gather_conv_bf16_wide_kernelILi2ELi4ELb1ELb1ELb1": (32, 8),
    "gather_conv_bf16_wide_kernelILi4ELi2ELb0ELb1ELb0": (32, 10),  # 512 x 128, two stages
    "gather_conv_bf16_wide_kernelILi4ELi2ELb1ELb1ELb0": (32, 10),
    "wgrad_bf16_wide_kernel": (32, 8),                              # 256 x 256, five-unit ring
    "wgrad_bf16_kernelILi4": (32, 12),
    "wgrad_bf16_kernelILi8": (16, 6),
}


def main(path):
    text = open(path).read()
    bad = []
    table, MFMA = (KERNELS_F32, MFMA_F32) if "conv_igemm" in path else (KERNELS, MFMA_BF16)
    for frag, (min_mfma, min_dma) in table.items():
        m = re.search(r"^(_ZN5mpgan\w*%s\w*):[^\n]*\n" % re.escape(frag), text, re.M)
        if not m:
            bad.append(f"{frag}: kernel not found in {path}")
            continue
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        # the K loop = the blocks LLVM annotates as members of the innermost loop that holds the MFMAs (block PLACEMENT is
        # not execution order: conditional issue blocks may sit behind the MFMAs they precede)
        blocks, cur = [], None
        for l in body:
            mm = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
            if mm:
                cur = {"label": mm.group(1), "head": mm.group(2), "lines": []}
                blocks.append(cur)
            elif cur is not None:
                if all(x.strip().startswith(";") for x in cur["lines"]) and l.strip().startswith(";") and "Loop" in l:
                    cur["head"] += " " + l            # (the annotation may continue on the next comment line)
                cur["lines"].append(l)
        loops = {}
        for b in blocks:
            ids = set(re.findall(r"(?:Header=|Parent Loop )(BB\d+_\d+)", b["head"]))
            if "Loop Header" in b["head"]:
                ids.add(b["label"].lstrip(".L"))
            for h in ids:
                loops.setdefault(h, []).append(b)
        best = None
        is_dma = lambda l: "global_load_lds_dwordx4" in l or ("buffer_load_dwordx4" in l and " lds" in l)
        for hdr, bl in loops.items():      # the smallest loop (with its nested ones) that holds both MFMAs and LDS-DMA loads
            lines = [l for b in bl for l in b["lines"]]
            if any(MFMA in l for l in lines) and any(is_dma(l) for l in lines) and \
                    (best is None or len(lines) < len(best)):
                best = lines
        if best is None:
            bad.append(f"{frag}: no loop with MFMAs found")
            continue
        loop = best
        if not any(re.search(r"s_waitcnt vmcnt\((?!0\))\d+\)", l) for l in body):
            bad.append(f"{frag}: no counted vmcnt wait in the kernel")
            continue
        n_mfma = sum(MFMA in l for l in loop)
        n_dma = sum("global_load_lds_dwordx4" in l or ("buffer_load_dwordx4" in l and " lds" in l) for l in loop)
        # (the kernel's own drain in front of its LAST tile carries a "; tail" comment)
        drains = [l.strip() for l in loop if "vmcnt(0)" in l and "; tail" not in l]
        scratch = [l.strip() for l in loop if l.strip().startswith("scratch_")]
        tail = text[m.end():text.index(".end_amdhsa_kernel", m.end())]
        priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", tail)
        if priv and int(priv.group(1)) != 0:
            bad.append(f"{frag}: {priv.group(1)} bytes of scratch per lane (a dynamically indexed kernel argument or a spill)")
        if drains:
            bad.append(f"{frag}: vmcnt(0) inside the K loop ({len(drains)}x): the LDS-DMA pipeline is serialised")
        if scratch:
            bad.append(f"{frag}: {len(scratch)} scratch accesses inside the K loop (register spill)")
        if n_mfma < min_mfma or n_dma < min_dma:
            bad.append(f"{frag}: K loop holds {n_mfma} MFMAs / {n_dma} LDS-DMA loads, expected >= {min_mfma} / {min_dma}")
        print(f"{frag:50s} K loop: {n_mfma} MFMA, {n_dma} LDS-DMA, {len(loop)} lines, no drain" if not drains else
              f"{frag:44s} DRAINED")
    if bad:
        print("\n".join(bad), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(max(main(p) for p in sys.argv[1:]))
