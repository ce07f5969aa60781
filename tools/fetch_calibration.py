#!/usr/bin/env python3
"""Report of tools/fetch_calibration.sh: counter values per measured kernel of tools/fetch_calibration.hip against its known byte count.
usage: tools/fetch_calibration.py <outdir>   (reads plain.csv and pass*/**/*_counter_collection.csv; prints markdown)"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    with open(f"{out}/plain.csv", newline="") as fh:
        plain = list(csv.DictReader(fh))
    # dispatches in launch order: [flush, measured] pairs; the flush and the control have the same kernel name, so go by order
    per_counter = {}
    for path in sorted(glob.glob(f"{out}/pass*/**/*_counter_collection.csv", recursive=True)):
        rows = defaultdict(lambda: defaultdict(float))
        names = {}
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                d = int(row["Dispatch_Id"])
                rows[d][row["Counter_Name"]] += float(row["Counter_Value"])
                names[d] = row["Kernel_Name"]
        order = [d for d in sorted(rows) if "cal_" in names[d]]  # (hipMemset's fill kernels are dispatches too)
        measured = order[1::2]  # every second dispatch is a measured kernel, the one before it the cache flush
        if len(measured) != len(plain):
            print(f"<!-- {path}: {len(order)} dispatches, expected {2 * len(plain)} of cal_* -->")
            continue
        for k, d in enumerate(measured):
            for c, v in rows[d].items():
                per_counter.setdefault(c, {})[k] = v
    counters = sorted(per_counter)
    print("| kernel | table MB | algorithmic bytes | GB/s (algorithmic) | " + " | ".join(counters) + " | FETCH_SIZE KiB x 1024 / bytes | RDREQ x 64 / bytes | L2 hit rate |")
    print("|---|---|---|---|" + "---|" * (len(counters) + 3))
    for k, row in enumerate(plain):
        b = float(row["algorithmic_bytes"])
        vals = [per_counter[c].get(k, float("nan")) for c in counters]
        fetch = per_counter.get("FETCH_SIZE", {}).get(k)
        rd = per_counter.get("TCC_EA0_RDREQ_sum", {}).get(k)
        hit, miss = per_counter.get("TCC_HIT_sum", {}).get(k), per_counter.get("TCC_MISS_sum", {}).get(k)
        print(f"| `{row['kernel']}` | {row['table_MB']} | {b:.4g} | {row['GB_per_s']} | " + " | ".join(f"{v:.5g}" for v in vals) +
              f" | {fetch * 1024 / b if fetch else float('nan'):.3f} | {rd * 64 / b if rd else float('nan'):.3f} | {hit / (hit + miss) if hit is not None and miss else float('nan'):.3f} |")


if __name__ == "__main__":
    main()
