#!/usr/bin/env python3
"""Per-kernel summary of an SQ counter pass (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE): matrix pipe
busy share of SIMD cycles = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), OTHER vector
instructions per MFMA (SQ_INSTS_VALU counts the MFMAs too).  Usage: tools/make_sq_summary.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import re
import sys


def main(src, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(src)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("mpgan::", "").split("(")[0]
        key = (name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[key].add(r["Dispatch_Id"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "launches", "mfma_busy_frac_of_simd_cycles", "valu_per_mfma",
                    "lds_bank_conflict_cycles", "waves", "mfma_insts", "valu_insts", "grbm_gui_active_sum8xcd"])
        for (name, grid), c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0)):
            if c.get("SQ_INSTS_MFMA", 0) < 1e6:
                continue
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else 0.0
            w.writerow([name, grid, len(launches[(name, grid)]), round(busy, 4),
                        round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"] - 1.0, 2), int(c.get("SQ_LDS_BANK_CONFLICT", 0)),
                        int(c.get("SQ_WAVES", 0)), int(c["SQ_INSTS_MFMA"]), int(c.get("SQ_INSTS_VALU", 0)), int(gui)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
