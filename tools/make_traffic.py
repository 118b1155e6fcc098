#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
`--kernel-trace --pmc <counter> --output-format csv`).  Per kernel and launch:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
FETCH_SIZE / WRITE_SIZE are in KiB; the factor 2 is the gfx950 half-count correction of
/opt/skills/guides/MI355X_MICROARCH.md ("HBM" section).  Usage:
    tools/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [main source]
(main source: the .hip file whose hash stamps the result; conv_igemm.hip by default, conv_bf16.hip for config C5)"""
import collections
import csv
import json
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("mpgan::", "").split("(")[0]
        tot[name] += float(r["Counter_Value"])
        launches[name].add(r["Dispatch_Id"])
    return {k: (tot[k] / max(len(launches[k]), 1), len(launches[k])) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    import bench
    # stamped with the kernel sources the passes ran on: bench.py quotes these bytes only for the same sources
    out = {"_source_sha256_16": bench.kernel_source_hash(sys.argv[4] if len(sys.argv) > 4 else "conv_igemm.hip")}
    for k, (f, n) in sorted(fetch.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        w = write.get(k, (0.0, 0))[0]
        out[k] = {"launches": n, "fetch_bytes_per_launch_corrected": 2 * f * 1024,
                  "write_bytes_per_launch": w * 1024, "hbm_bytes_per_launch": (2 * f + w) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k in [k for k in out if not k.startswith('_')][:8]:
        print(f"{k:70s} {out[k]['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x{out[k]['launches']}")


if __name__ == "__main__":
    main()
