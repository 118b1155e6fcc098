#!/usr/bin/env python3
"""Build-time check of the bf16 kernels' K loops (run by __graft_entry__.build()): the LDS-DMA pipeline only
works if the loop body waits with the COUNTED vmcnt it was written with.  hipcc inserts `s_waitcnt vmcnt(0)`
in front of any LDS access it thinks may alias an LDS-DMA in flight, silently serialising the pipeline; this
script reads the device assembly hipcc leaves beside the object (-save-temps=obj) and fails if the innermost
loop of a pipelined kernel holds a vmcnt(0), scratch traffic, or fewer MFMAs / DMA loads than expected."""
import re
import sys

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
    "wgrad_bf16_kernelILi4": (32, 12),
    "wgrad_bf16_kernelILi8": (16, 6),
}


def main(path):
    text = open(path).read()
    bad = []
    for frag, (min_mfma, min_dma) in KERNELS.items():
        m = re.search(r"^(_ZN5mpgan\w*%s\w*):[^\n]*\n" % re.escape(frag), text, re.M)
        if not m:
            bad.append(f"{frag}: kernel not found in {path}")
            continue
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        # the K loop = the span between the counted wait and the last MFMA that follows it
        idx = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\((?!0\))\d+\)", l)]
        if not idx:
            bad.append(f"{frag}: no counted vmcnt wait in the kernel")
            continue
        start = idx[0]
        mf = [i for i, l in enumerate(body) if "v_mfma_f32_32x32x16_bf16" in l and i > start]
        end = mf[-1] if mf else start
        loop = body[start:end + 1]
        n_mfma = sum("v_mfma_f32_32x32x16_bf16" in l for l in loop)
        n_dma = sum("global_load_lds_dwordx4" in l for l in loop)
        # (the kernel's own drain in front of its LAST tile carries a "; tail" comment)
        drains = [l.strip() for l in loop if "vmcnt(0)" in l and "; tail" not in l]
        scratch = [l.strip() for l in loop if l.strip().startswith("scratch_")]
        if drains:
            bad.append(f"{frag}: vmcnt(0) inside the K loop ({len(drains)}x): the LDS-DMA pipeline is serialised")
        if scratch:
            bad.append(f"{frag}: {len(scratch)} scratch accesses inside the K loop (register spill)")
        if n_mfma < min_mfma or n_dma < min_dma:
            bad.append(f"{frag}: K loop holds {n_mfma} MFMAs / {n_dma} LDS-DMA loads, expected >= {min_mfma} / {min_dma}")
        print(f"{frag:44s} K loop: {n_mfma} MFMA, {n_dma} LDS-DMA, {len(loop)} lines, no drain" if not drains else
              f"{frag:44s} DRAINED")
    if bad:
        print("\n".join(bad), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
