#!/usr/bin/env python3
"""Aggregate the per-pass rocprofv3 --pmc outputs of tools/pmc_passes.sh into one small CSV.

usage: tools/pmc_aggregate.py <pmc outdir> <kernel substring> > profiles/rNN_pmc_render_kernel.csv

For every counter: mean over the dispatches of kernels whose name contains the substring (counter values of one
dispatch are summed over the rows rocprofv3 emits for it, e.g. one row per XCD/instance)."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    outdir, needle = sys.argv[1], sys.argv[2]
    per_dispatch = defaultdict(float)
    for path in sorted(glob.glob(f"{outdir}/pass*/**/*_counter_collection.csv", recursive=True)):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if needle not in row["Kernel_Name"]:
                    continue
                per_dispatch[(path, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    sums, counts = defaultdict(float), defaultdict(int)
    for (_, _, name), v in per_dispatch.items():
        sums[name] += v
        counts[name] += 1
    print("counter,mean_per_launch,launches")
    for name in sums:
        print(f"{name},{sums[name] / counts[name]:.6g},{counts[name]}")


if __name__ == "__main__":
    main()
